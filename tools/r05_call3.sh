#!/usr/bin/env bash
# Round 5, GPU call 3: (1) the tests touched since call 2 (GroupNorm algorithm flag, GEMM epilogue, conv, peaked / full-map / fusion
# parity), (2) same-box A/B of the 3x3 convolution forms: raster (rounds 2-4) / halo 8x16x64 (default) / halo 8x16x160 experiment
# -- kernel table and whole step.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c3
timeout 900 python -m pytest tests -m gpu -q -rP --maxfail=8 -k "groupnorm or gemm or lora_linear or conv3x3 or peaked or full_maps or three_sd15_layers or update_quasi_newton or softmax_rows or materialised or region_attention or feed_forward or graft_smoke" > "$O/${TAG}_gpu_tests_changed.log" 2>&1
echo "pytest rc=$?"; tail -6 "$O/${TAG}_gpu_tests_changed.log" | cut -c1-200
grep -E "^\[parity\].*(peaked|full-map|FUSED WEIGHT|lbfgs\[|free-running final)" "$O/${TAG}_gpu_tests_changed.log" | cut -c1-520 | sort -u | head -24
grep -E "^smoke:" "$O/${TAG}_gpu_tests_changed.log" | head -2
echo "== conv kernel table, three builds"
for lib in "" "_variants/libmos_hip_raster_conv.so" "_variants/libmos_hip_halo160.so"; do
  echo "-- MOS_HIP_LIB=[$lib]"
  MOS_HIP_LIB="${lib:+$ROOT/$lib}" timeout 200 python tools/bench_kernels.py --only conv --iters 30 --ref 0 2>&1 | grep -E "^B[24] " | cut -c1-120
done > "$O/${TAG}_kernel_bench_conv_forms.txt" 2>&1
cat "$O/${TAG}_kernel_bench_conv_forms.txt"
echo "== whole step / sample, three builds, interleaved twice"
for rep in 1 2; do
for lib in "" "_variants/libmos_hip_raster_conv.so" "_variants/libmos_hip_halo160.so"; do
  MOS_HIP_LIB="${lib:+$ROOT/$lib}" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=[k for k in d['dominant_kernels_by_name'] if k['kernel']=='conv3x3'][0]; r=[k for k in d['regional']['dominant_kernels_by_name'] if k['kernel']=='conv3x3'][0]
print('[$lib] train', d['value'], 'img/s', d['ms_per_step'], 'ms; conv3x3', c['ms'], 'ms/step', c['frac_of_mfma_peak'], '; regional image', d['regional_ms_image'], 'latent', d['regional_ms_latent'], 'conv3x3', r['ms'], 'ms/sample')"
done; done 2>&1 | tee "$O/${TAG}_ab_same_box_conv_forms.txt"
