#!/usr/bin/env bash
# Round 5, GPU call 1: first device run of the merged next/wide-tiles code (256-row tiles, halo conv, L-BFGS history kernels).
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
T="$O/r05c1_wide_tiles.txt"; : > "$T"
for knob in "" "MOS_CONV_TILE=256128 MOS_GEMM_TILE=256128" "MOS_CONV_TILE=25664 MOS_GEMM_TILE=25664" "MOS_CONV_HALO=864" "MOS_CONV_HALO=8128" "MOS_CONV_HALO=1664" "MOS_CONV_HALO=16128"; do
  echo "== [$knob] primitives" >> "$T"
  env $knob timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "conv3x3 or lora_ or gemm_" 2>&1 | tail -3 >> "$T"
  echo "== [$knob] kernel bench" >> "$T"
  env $knob timeout 200 python tools/bench_kernels.py --only conv,gemm,ff --iters 30 --ref 0 2>&1 | grep -vE "^JSON" | cut -c1-200 >> "$T"
done
tail -60 "$T"
echo "== L-BFGS history kernels"
timeout 120 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "lbfgs_history" 2>&1 | tail -2
for knob in "MOS_LBFGS_FUSED=1" "MOS_LBFGS_FUSED=1 MOS_LBFGS_HIST=f32" "MOS_LBFGS_FUSED=0"; do
  env $knob timeout 200 python bench.py --mode fusion --concepts 14 --steps 1 --warmup 0 --no-cpu-baseline 2>"$O/r05c1_fusion.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$knob]', d['value'], d.get('solve_seconds_last_pass'), d.get('stage_seconds_last_pass'))"
done 2>&1 | tee "$O/r05c1_fusion_ab.txt"
tail -3 "$O/r05c1_fusion.err"
echo "== whole-step A/B"
for knob in "" "MOS_CONV_TILE=256128 MOS_GEMM_TILE=256128"; do
  env $knob timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$knob]', d['value'], d['ms_per_step'], d.get('regional',{}).get('value_ms_image'), d.get('regional',{}).get('value_ms_latent'))"
done 2>&1 | tee "$O/r05c1_step_ab.txt"
