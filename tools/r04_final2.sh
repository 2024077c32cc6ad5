#!/usr/bin/env bash
# after the final evidence run: the whole GPU suite again (one fix in the layout checks), threaded fusion solves
set -u
TAG="${1:-r04h}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== full GPU test suite (-x, as the driver runs it)"
timeout 1100 python -m pytest tests -m gpu -q -x --durations=5 > "$O/${TAG}_gpu_tests.log" 2>&1
echo "rc=$?"; tail -9 "$O/${TAG}_gpu_tests.log" | cut -c1-200
echo "== fusion, sequential solves (1 pass)"
MOS_FUSION_THREADS=1 timeout 300 python bench.py --mode fusion --concepts 14 --steps 1 --warmup 0 --no-cpu-baseline > "$O/${TAG}_bench_fusion_threads1.json" 2> "$O/${TAG}_bench_fusion_threads1.err"
grep "fusion pass" "$O/${TAG}_bench_fusion_threads1.err"
echo "== fusion, 3 solver threads (default), then 6"
timeout 420 python bench.py --mode fusion --concepts 14 --steps 2 --warmup 1 > "$O/${TAG}_bench_fusion.json" 2> "$O/${TAG}_bench_fusion.err"
echo "rc=$?"; grep "fusion pass" "$O/${TAG}_bench_fusion.err"; cut -c1-200 "$O/${TAG}_bench_fusion.json"
MOS_FUSION_THREADS=6 timeout 300 python bench.py --mode fusion --concepts 14 --steps 1 --warmup 0 --no-cpu-baseline > "$O/${TAG}_bench_fusion_threads6.json" 2> "$O/${TAG}_bench_fusion_threads6.err"
grep "fusion pass" "$O/${TAG}_bench_fusion_threads6.err"
