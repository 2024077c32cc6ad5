#!/usr/bin/env bash
# Round-4, fifth GPU call: pipelined attention forward (stage-ahead operand reads) and the slot-interleaved dK/dV.
set -u
TAG="${1:-r04e}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== primitives (attention)"
timeout 400 python -m pytest tests/test_gpu_primitives.py -m gpu -q -s -k "attention" > "$O/${TAG}_primitives.log" 2>&1
echo "rc=$?"; tail -3 "$O/${TAG}_primitives.log"; grep -E "^FAILED|^E  " "$O/${TAG}_primitives.log" | head -10
grep "pipelined dK/dV.*bit-identical\|pipelined dK/dV.*dk vs emulation" "$O/${TAG}_primitives.log" | head -10
echo "== attention kernel bench: MOS_ATTN_PIPE=1 (forward + dK/dV pipelined), 2 (forward only), 0 (neither)"
for p in 1 2 0; do
  MOS_ATTN_PIPE=$p timeout 200 python tools/bench_kernels.py --only attn --iters 20 > "$O/${TAG}_kernel_bench_attn_pipe${p}.txt" 2>&1
  echo "-- MOS_ATTN_PIPE=$p"; grep -E "^attn_(fwd|bwd_dkdv|bwd_dq) f16 d40 B(2|4) H8 Nq(4096|6144) Nkv(4096|6144)" "$O/${TAG}_kernel_bench_attn_pipe${p}.txt"
done
echo "== same-box A/B, train half (10 steps)"
timeout 400 python tools/ab_switches.py --half train --kernels conv3x3,attn_fwd,attn_bwd_dkdv,attn_bwd_dq \
  "" "MOS_ATTN_PIPE=0" "MOS_ATTN_PIPE=2" > "$O/${TAG}_ab_train.txt" 2>&1
cut -c1-170 "$O/${TAG}_ab_train.txt"
echo "== same-box A/B, regional half (5 timed samples per row)"
timeout 700 python tools/ab_switches.py --half regional --steps 5 --timeout 300 --kernels conv3x3,attn_fwd,gemm_nt,groupnorm_fused \
  "" "MOS_ATTN_PIPE=0" "MOS_CONV_SPLIT_TILE=128" "MOS_CONV_SPLIT_TILE=128 MOS_ATTN_PIPE=0" > "$O/${TAG}_ab_regional.txt" 2>&1
cut -c1-170 "$O/${TAG}_ab_regional.txt"
