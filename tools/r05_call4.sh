#!/usr/bin/env bash
# Round 5, GPU call 4: halo convolution with the activation fragments read one step ahead (variant build) -- parity, kernel table,
# whole step, against the default library on the same box.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c4
V="$ROOT/_variants/libmos_hip_halo_prefetch.so"
MOS_HIP_LIB="$V" timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "conv3x3" 2>&1 | tail -3
for lib in "" "$V"; do
  echo "-- MOS_HIP_LIB=[$lib]"
  MOS_HIP_LIB="$lib" timeout 200 python tools/bench_kernels.py --only conv --iters 30 --ref 0 2>&1 | grep -E "^B[24] " | cut -c1-120
done > "$O/${TAG}_kernel_bench_conv_prefetch.txt" 2>&1
cat "$O/${TAG}_kernel_bench_conv_prefetch.txt"
for rep in 1 2; do
for lib in "" "$V"; do
  MOS_HIP_LIB="$lib" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=[k for k in d['dominant_kernels_by_name'] if k['kernel']=='conv3x3'][0]; r=[k for k in d['regional']['dominant_kernels_by_name'] if k['kernel']=='conv3x3'][0]
print('[$lib] train', d['value'], 'img/s', d['ms_per_step'], 'ms; conv3x3', c['ms'], 'ms/step', c['frac_of_mfma_peak'], '; regional image', d['regional_ms_image'], 'latent', d['regional_ms_latent'], 'conv3x3', r['ms'], 'ms/sample')"
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_conv_prefetch.txt"
