#!/usr/bin/env bash
# Round 5, GPU call 6: the token reductions of the LoRA factor gradients batched into one launch per rank class
# (mos_lora_grad_all): parity (bit-identical gradients), training-side e2e tests, same-box A/B of the whole step.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c6
timeout 900 python -m pytest tests -m gpu -q --maxfail=6 -k "lora or train or hipgraph or smoke or rccl or engine" > "$O/${TAG}_gpu_tests_training_side.log" 2>&1
echo "pytest rc=$?"; tail -8 "$O/${TAG}_gpu_tests_training_side.log" | cut -c1-200
for rep in 1 2; do
for knob in "MOS_LORA_GRAD_BATCH=1" "MOS_LORA_GRAD_BATCH=0"; do
  env $knob timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regional 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=[k for k in d['dominant_kernels_by_name'] if k['kernel'].startswith('lora_grad')]
print('[$knob] train', d['value'], 'img/s', d['ms_per_step'], 'ms;', [(k['kernel'], k['ms'], k['launches'], k['gbps']) for k in c])"
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_lora_grad_batch.txt"
