#!/usr/bin/env bash
# LDS-DMA staging: parity of the conv / GEMM primitives under the switches, kernel A/B timings, whole-step A/B -- one box.
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
t() { timeout 400 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x --timeout 300 -k "$1" 2>&1 | tail -2; }
step() { timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-regional 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms/step; lib', d['library_kernel_ms_per_step'])"; }
echo "== parity conv (DMA=1 default)"; t "conv3x3"
echo "== parity gemm (DMA=1)"; MOS_GEMM_DMA=1 t "linear or lora"
echo "== parity conv DMA=2 (256x128 tiles), big case"; MOS_CONV_DMA=2 t "conv3x3 and 512-500"
for v in 0 1 2; do
  echo "== conv timings MOS_CONV_DMA=$v"
  MOS_CONV_DMA=$v timeout 300 python tools/bench_kernels.py --only conv --ref 0 --iters 12 2>/dev/null | grep -E "^B[0-9]"
done
for v in 0 1; do
  echo "== gemm timings MOS_GEMM_DMA=$v"
  MOS_GEMM_DMA=$v timeout 300 python tools/bench_kernels.py --only gemm --ref 0 --iters 20 2>/dev/null | grep -E "^gemm_nt" | cut -c1-120
done
echo "== step conv0 gemm0"; MOS_CONV_DMA=0 MOS_GEMM_DMA=0 step
echo "== step conv1 gemm0"; MOS_CONV_DMA=1 MOS_GEMM_DMA=0 step
echo "== step conv1 gemm1"; MOS_CONV_DMA=1 MOS_GEMM_DMA=1 step
echo "== step conv2 gemm0"; MOS_CONV_DMA=2 MOS_GEMM_DMA=0 step
