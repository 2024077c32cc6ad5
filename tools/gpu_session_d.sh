#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (new: causal d64, softmax/vae attention, groupnorm nhwc)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "causal or softmax or groupnorm or attention" 2>&1 | tail -4
echo "== e2e: smoke + sd15 train step"; timeout 900 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --timeout 900 -k "smoke or sd15_fp16_train" > $O/r02_e2e_d.log 2>&1; grep -E "parity|smoke:|passed|failed" $O/r02_e2e_d.log | cut -c1-500
echo "== bench train"; timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-regional > $O/r02_bench_train_d.json 2> $O/r02_bench_train_d.err
python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/r02_bench_train_d.json'))
    print(d['value'],'img/s',d['ms_per_step'],'ms/step lib',d['library_kernel_ms_per_step'])
    for k in d['kernels'][:14]: print('   ',k)
except Exception as e:
    print('bench failed',e); print(open('gpurun_out/r02_bench_train_d.err').read()[-3000:])
PY
echo "== steady-state kernel breakdown (torch profiler, eager)"; timeout 600 python tools/profile_step.py --mode train > $O/r02_step_breakdown_train.txt 2>$O/r02_step_breakdown_train.err; head -95 $O/r02_step_breakdown_train.txt | cut -c1-190
