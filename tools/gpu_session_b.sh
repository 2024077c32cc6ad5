#!/usr/bin/env bash
# round-2 GPU session B: new elementwise kernels + split lora_grad (parity), e2e parity with them in the path, bench NCHW vs NHWC
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives"; timeout 900 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "lora or linear or pack or groupnorm or layernorm or geglu" > $O/r02_prim_b.log 2>&1; tail -5 $O/r02_prim_b.log
echo "== gemm/lora micro-benchmark"; timeout 300 python tools/bench_kernels.py --only gemm --iters 20 2>&1 | grep -v JSON > $O/r02_kb_gemm2.txt; grep -E "lora_grad|fused" $O/r02_kb_gemm2.txt
echo "== e2e parity (subset)"; timeout 1500 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --timeout 900 -k "hot_path or smoke or sd15_fp16_train or hipgraph_step" > $O/r02_e2e_c.log 2>&1; grep -E "parity|smoke:|passed|failed" $O/r02_e2e_c.log | cut -c1-600
for cl in 0 1; do
  echo "== bench train channels_last=$cl"
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-regional --channels-last $cl > $O/r02_bench_train_cl$cl.json 2> $O/r02_bench_train_cl$cl.err
  python - <<PY
import json
try:
    d=json.load(open('$O/r02_bench_train_cl$cl.json'))
    print(d['value'],'img/s',d['ms_per_step'],'ms/step lib',d['library_kernel_ms_per_step'],'attn_path',d['attention_path'])
    for k in d['kernels'][:14]: print('   ',k)
except Exception as e:
    print('bench failed',e); print(open('$O/r02_bench_train_cl$cl.err').read()[-3000:])
PY
  echo "== bench regional channels_last=$cl"
  timeout 600 python bench.py --mode regional --steps 3 --warmup 1 --no-cpu-baseline --channels-last $cl > $O/r02_bench_regional_cl$cl.json 2> $O/r02_bench_regional_cl$cl.err
  python - <<PY
import json
try:
    d=json.load(open('$O/r02_bench_regional_cl$cl.json'))
    print(d['value'],'ms/sample lib',d['library_kernel_ms_per_sample'])
    for k in d['kernels'][:8]: print('   ',k)
except Exception as e:
    print('bench failed',e); print(open('$O/r02_bench_regional_cl$cl.err').read()[-3000:])
PY
done
