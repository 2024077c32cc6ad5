#!/usr/bin/env bash
# Round 5, GPU call 13: (1) the conv parity cases (three new ones: the 16x16x128 / 32-channel tile ragged, upsampled) on the library
# as built; (2) variant build -DMOS_DMA_SOFF (csrc/build.sh MOS_CONV_FLAGS / MOS_GEMM_FLAGS): the LDS-DMA ring of the LoRA GEMM and of
# the raster / split-K convolution with the K offset in the scalar-offset operand and zero-record descriptors past the end
# (VALU per K step 23 -> 9, 19 -> 9, 18 -> 12): parity cases, then the whole step on the same box.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c13
V="$ROOT/_variants/libmos_hip_soff.so"
echo "== conv parity [as built]"
timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "conv3x3" 2>&1 | tail -3
echo "== gemm / lora / conv parity [soff]"
MOS_HIP_LIB="$V" timeout 400 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "conv3x3 or lora or gemm or geglu or linear" 2>&1 | tail -3
for rep in 1 2 3; do
for name in base soff; do
  L=""; [ "$name" = soff ] && L="$V"
  MOS_HIP_LIB="$L" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=[k for k in d['dominant_kernels_by_name'] if k['kernel'].startswith(('conv3x3','gemm_nt','lora_linear'))]
r=d.get('regional',{})
print('[$name] train', d['value'], 'img/s', d['ms_per_step'], 'ms; regional', r.get('value_ms_latent'), '/', r.get('value_ms_image'), 'ms;', [(k['kernel'], k['ms']) for k in c])"
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_dma_scalar_offset.txt"
