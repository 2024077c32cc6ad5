#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (groupnorm)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "groupnorm" 2>&1 | tail -3
echo "== rocprof bench"; bash tools/rocprof_bench.sh r02a --no-regional 2>&1 | tail -5
python - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/r02a_rocprofv3_kernel_stats_bench_train.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:70]:
    n=re.sub(r'void at::native::','',r['Name'])[:140]
    print(f"{float(r['TotalDurationNs'])/1e6:8.3f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:8.1f} us  {n}")
PY
