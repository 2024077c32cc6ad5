#!/usr/bin/env bash
# tap masks in the convolution loop + VGPR-form MFMA accumulators for the GEMM / conv translation units: validate, A/B against the
# previous library on the same box (_variants/libmos_hip_before.so), then -- only if green -- the whole evidence run again.
set -u
TAG="${1:-r04i}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== primitives (GEMM family, convolutions)"
timeout 500 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "conv3x3 or lora_ or gemm_" > "$O/${TAG}_primitives.log" 2>&1
rc1=$?; echo "rc=$rc1"; tail -2 "$O/${TAG}_primitives.log"
echo "== end to end (smoke, training parity, graph)"
timeout 500 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -x -s \
  -k "graft_smoke or training_steps_match or train_step_through_vae or hipgraph_step_equals or pipeline_call_equals" > "$O/${TAG}_e2e.log" 2>&1
rc2=$?; echo "rc=$rc2"; grep -E "^\[parity\]|passed|failed" "$O/${TAG}_e2e.log" | cut -c1-250 | tail -6
if [ $rc1 -ne 0 ] || [ $rc2 -ne 0 ]; then echo "VALIDATION FAILED: stopping"; exit 1; fi
echo "== same-box A/B against the previous library"
timeout 400 python tools/ab_switches.py --half train --kernels conv3x3,gemm_nt,lora_grad \
  "" "MOS_HIP_LIB=$ROOT/_variants/libmos_hip_before.so" "" "MOS_HIP_LIB=$ROOT/_variants/libmos_hip_before.so" > "$O/${TAG}_ab_train.txt" 2>&1
cut -c1-150 "$O/${TAG}_ab_train.txt"
timeout 500 python tools/ab_switches.py --half regional --steps 3 --timeout 300 --kernels conv3x3,gemm_nt \
  "" "MOS_HIP_LIB=$ROOT/_variants/libmos_hip_before.so" > "$O/${TAG}_ab_regional.txt" 2>&1
cut -c1-150 "$O/${TAG}_ab_regional.txt"
echo "== kernel bench: convolutions"
timeout 200 python tools/bench_kernels.py --only conv --iters 30 --ref 0 > "$O/${TAG}_kernel_bench_conv.txt" 2>&1
grep -E "^B[24] " "$O/${TAG}_kernel_bench_conv.txt" | head -24
echo "================ evidence run on the validated tree"
TESTS_TIMEOUT=900 bash tools/final_gpu_run.sh r04v2
