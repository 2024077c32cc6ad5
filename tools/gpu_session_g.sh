#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (conv3x3)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "conv3x3" 2>&1 | tail -4
echo "== conv micro-benchmark vs MIOpen"; timeout 600 python tools/bench_kernels.py --only conv --iters 10 2>&1 | grep -v JSON | grep -E "conv3x3|^B[0-9]" > $O/r02_kb_conv.txt; cat $O/r02_kb_conv.txt
step() { timeout 600 python bench.py --no-cpu-baseline --no-regional --steps 16 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms; lib', d['library_kernel_ms_per_step'])"; }
echo "== bench train, conv3x3 on"; step
echo "== bench train, conv3x3 off (MIOpen)"; MOS_CONV3X3=0 step
echo "== e2e: smoke + sd15 train step (conv on)"; timeout 900 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --timeout 900 -k "smoke or sd15_fp16_train" > $O/r02_e2e_g.log 2>&1; grep -E "parity|smoke:|passed|failed" $O/r02_e2e_g.log | cut -c1-500
echo "== regional, conv on / off"; for v in 1 0; do MOS_CONV3X3=$v timeout 600 python bench.py --mode regional --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'ms/sample; lib', d['library_kernel_ms_per_sample'])"; done
