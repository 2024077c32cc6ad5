#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out
reg() { timeout 600 python bench.py --mode regional --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'ms/sample; lib', d['library_kernel_ms_per_sample']); [print('   ',k) for k in d['kernels'][:10]]"; }
echo "== regional: kernels under no_grad"; reg
echo "== regional: conv off"; MOS_CONV3X3=0 reg | head -1
echo "== e2e (sampling + fusion)"; timeout 1500 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --timeout 900 -k "fp16_pipeline or written_out or hipgraph_regional or fusion" > $O/r02_e2e_j.log 2>&1; grep -E "parity\] (edlora|regional|EDLoRA|Regionally|hipgraph|fusion (text|cross|spatial):)|passed|failed" $O/r02_e2e_j.log | cut -c1-420
