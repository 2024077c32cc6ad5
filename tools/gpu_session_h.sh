#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (conv3x3)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -k "conv3x3" 2>&1 | tail -3
step() { timeout 600 python bench.py --no-cpu-baseline --no-regional --steps 16 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms; lib', d['library_kernel_ms_per_step'])"; }
echo "== bench train, conv3x3 >= 8192 px"; step
echo "== bench train, conv3x3 >= 4096 px"; MOS_CONV3X3_MIN_PIXELS=4096 step
echo "== bench train, conv3x3 off (MIOpen)"; MOS_CONV3X3=0 step
echo "== regional, conv >=8192 / off"; for v in 1 0; do MOS_CONV3X3=$v timeout 600 python bench.py --mode regional --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'ms/sample; lib', d['library_kernel_ms_per_sample'])"; done
