#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (lora/linear)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "lora or linear or pack or softmax" 2>&1 | tail -3
echo "== gemm micro-benchmark (32-row tiles for small grids) + hipBLASLt reference"; timeout 300 python tools/bench_kernels.py --only gemm --iters 20 2>&1 | grep -v JSON | grep -E "gemm_nt|GEMM|^M[0-9]" > $O/r02_kb_gemm4.txt; cat $O/r02_kb_gemm4.txt
echo "== bench train"; for i in 1 2; do timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-regional 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms')"; done
echo "== fusion bench (configs[3])"; timeout 1200 python bench.py --mode fusion --steps 1 --warmup 0 > $O/r02_bench_fusion.json 2> $O/r02_bench_fusion.err; tail -5 $O/r02_bench_fusion.err; cut -c1-1500 $O/r02_bench_fusion.json
