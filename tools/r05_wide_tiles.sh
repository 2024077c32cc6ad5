#!/usr/bin/env bash
# Branch next/wide-tiles: 256-row workgroup tiles for the convolution / GEMM kernels behind MOS_CONV_TILE / MOS_GEMM_TILE
# (256128 = 256 x 128 ring of 3, 25664 = 256 x 64, conv only: 256256). Compiled and ISA-checked without a GPU in round 4
# MOS_CONV_HALO (864 / 8128 / 1664 / 16128 = TH x BN): the halo-staged convolution (conv3x3_halo_kernel: index logic checked with the
# lane-level model tools/next/sim_conv_halo.py; the wait / barrier protocol is reasoned, not run).
# (no scratch; 207 / 306 VGPRs; 64 MFMAs per K tile against 12 LDS-DMA pieces and 24 ds_read_b128 per wave, the 128 x 128
# form: 32 / 8 / 16). First thing to run on the device:  parity of the conv / GEMM primitives with the knob set, then the kernel bench.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
for knob in "" "MOS_CONV_TILE=256128 MOS_GEMM_TILE=256128" "MOS_CONV_TILE=25664 MOS_GEMM_TILE=25664" "MOS_CONV_HALO=864" "MOS_CONV_HALO=8128" "MOS_CONV_HALO=1664" "MOS_CONV_HALO=16128"; do
  echo "== [$knob] primitives"
  env $knob timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "conv3x3 or lora_ or gemm_" 2>&1 | tail -2
  echo "== [$knob] kernel bench"
  env $knob timeout 200 python tools/bench_kernels.py --only conv,ff --iters 30 --ref 0 2>&1 | grep -E "^B[24] |^M" | head -40
done > "$O/r05_wide_tiles.txt" 2>&1
cat "$O/r05_wide_tiles.txt"
echo "== L-BFGS history kernels: primitive parity, then one fusion pass with / without them (MOS_LBFGS_FUSED)"
timeout 120 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "lbfgs_history" 2>&1 | tail -2
for knob in "" "MOS_LBFGS_HIST=f32" "MOS_LBFGS_FUSED=0"; do
  env $knob timeout 150 python bench.py --mode fusion --concepts 14 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$knob]', d['value'], d.get('solve_seconds_last_pass'))"
done
