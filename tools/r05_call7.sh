#!/usr/bin/env bash
# Round 5, GPU call 7: tuning of the batched LoRA gradient reduction (variant builds): unroll depth, workgroups aimed at per group.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c7
for rep in 1 2; do
for lib in "" u8 wg512 wg256 u8wg256; do
  L=""; [ -n "$lib" ] && L="$ROOT/_variants/libmos_hip_grad_$lib.so"
  MOS_HIP_LIB="$L" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regional 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=[k for k in d['dominant_kernels_by_name'] if k['kernel'].startswith('lora_grad')]
print('[$lib] train', d['value'], 'img/s', d['ms_per_step'], 'ms;', [(k['kernel'], k['ms'], k['launches'], k['gbps']) for k in c])"
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_lora_grad_tuning.txt"
MOS_HIP_LIB="$ROOT/_variants/libmos_hip_grad_u8wg256.so" timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "lora" 2>&1 | tail -2
