#!/usr/bin/env bash
# Round-4 kernel validation + same-box A/Bs: the one-launch GroupNorm column kernel, the GEMM residual / GEGLU epilogues, the
# conv dispatch threshold at 512x768 and the batched time-embedding projections.       bash tools/r04_validate_kernels.sh [tag]
set -u
TAG="${1:-r04b}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== primitives (GroupNorm incl. column kernel, GEMM epilogues, LoRA linear, conv incl. split-K)"
timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -s -k "groupnorm or gemm_residual or gemm_geglu or lora_linear or conv3x3" \
  > "$O/${TAG}_primitives.log" 2>&1
echo "rc=$?"; tail -3 "$O/${TAG}_primitives.log"
grep -c "bit-identical.*True" "$O/${TAG}_primitives.log"; grep "bit-identical.*False" "$O/${TAG}_primitives.log" | head -8
grep -E "FAILED|Error|assert" "$O/${TAG}_primitives.log" | head -10
echo "== kernel bench: feed-forward and GroupNorm paths"
timeout 400 python tools/bench_kernels.py --only ff,gn,conv --iters 30 > "$O/${TAG}_kernel_bench_ff_gn.txt" 2>&1
echo "rc=$?"; grep -v "^JSON\|^kernel  " "$O/${TAG}_kernel_bench_ff_gn.txt" | head -75
echo "== end to end (training side, graph, smoke, sampling written-out loop)"
timeout 420 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --durations=6 \
  -k "graft_smoke or training_steps_match or hipgraph_step_equals or pipeline_call_equals or hipgraph_regional" \
  > "$O/${TAG}_e2e.log" 2>&1
echo "rc=$?"; grep -E "^\[parity\]|passed|failed|Error" "$O/${TAG}_e2e.log" | cut -c1-300 | tail -10
echo "== same-box A/B, train half"
timeout 600 python tools/ab_switches.py --half train --kernels conv3x3,gemm_nt,groupnorm_fused,groupnorm_apply,groupnorm_stats,groupnorm_bwd_fused,geglu_fwd \
  "" "MOS_GN_FUSED=0" "MOS_GN_FUSED=2" "MOS_FF2_OWN=0 MOS_GEMM_RESIDUAL=0" "MOS_CONV3X3_MIN_PIXELS=0" "MOS_CONV3X3_MIN_PIXELS=0 MOS_CONV_SPLITK=0" \
  "MOS_BATCH_TEMB=1" > "$O/${TAG}_ab_train.txt" 2>&1
cut -c1-200 "$O/${TAG}_ab_train.txt"
echo "== same-box A/B, regional half"
timeout 1000 python tools/ab_switches.py --half regional --timeout 300 --kernels conv3x3,gemm_nt,groupnorm_fused,groupnorm_apply,groupnorm_stats,geglu_fwd,attn_fwd \
  "" "MOS_GN_FUSED=0" "MOS_GN_FUSED=2" "MOS_FF_GEGLU=0" "MOS_FF2_OWN=0 MOS_GEMM_RESIDUAL=0" "MOS_CONV3X3_MIN_PIXELS=3072" "MOS_CONV3X3_MIN_PIXELS=0" "MOS_BATCH_TEMB=1" \
  > "$O/${TAG}_ab_regional.txt" 2>&1
cut -c1-200 "$O/${TAG}_ab_regional.txt"
