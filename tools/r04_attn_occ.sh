#!/usr/bin/env bash
# one wave per SIMD for the slot-interleaved attention kernels? (MOS_ATTN_PIPE_OCC=1 forces one workgroup per CU)
set -u
TAG="${1:-r04f}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
for occ in 0 1; do
  MOS_ATTN_PIPE=1 MOS_ATTN_PIPE_OCC=$occ timeout 200 python tools/bench_kernels.py --only attn --iters 20 > "$O/${TAG}_kernel_bench_attn_occ${occ}.txt" 2>&1
  echo "-- MOS_ATTN_PIPE=1 MOS_ATTN_PIPE_OCC=$occ"; grep -E "^attn_(fwd|bwd_dkdv) f16 d40 B(2|4) H8 Nq(4096|6144) Nkv(4096|6144)" "$O/${TAG}_kernel_bench_attn_occ${occ}.txt"
done
