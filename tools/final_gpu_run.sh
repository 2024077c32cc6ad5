#!/usr/bin/env bash
# One GPU-box pass that regenerates the evidence under profiles/: parity logs, bench lines, rocprofv3 kernel stats.
# Run from the repo root on the GPU box:  bash tools/final_gpu_run.sh <tag>     (outputs -> gpurun_out/<tag>_*)
set -u
TAG="${1:-r01}"
ROOT="$(pwd)"
OUT="${ROOT}/gpurun_out"
mkdir -p "${OUT}"
timeout 420 python -m pytest tests/test_gpu_primitives.py -m gpu -q 2>&1 | tail -4 > "${OUT}/${TAG}_gpu_primitive_parity.log"
timeout 420 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s 2>&1 | grep -E "parity|passed|failed|Error" > "${OUT}/${TAG}_gpu_end_to_end_parity.log"
timeout 420 python bench.py > "${OUT}/${TAG}_bench_train_n1.json" 2> "${OUT}/${TAG}_bench_train_n1.err"
timeout 300 python bench.py --mode regional > "${OUT}/${TAG}_bench_regional_n1.json" 2> "${OUT}/${TAG}_bench_regional_n1.err"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python "${ROOT}/bench.py" --steps 3 --warmup 2 --no-cpu-baseline \
    > "${OUT}/${TAG}_bench_train_under_rocprof.json" 2> "${OUT}/${TAG}_rocprof.err"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -60 "$f" > "${OUT}/${TAG}_rocprofv3_kernel_stats_bench_train.csv"
cd "${ROOT}"
tail -2 "${OUT}/${TAG}_gpu_primitive_parity.log"; tail -3 "${OUT}/${TAG}_gpu_end_to_end_parity.log"
cut -c1-260 "${OUT}/${TAG}_bench_train_n1.json"; cut -c1-200 "${OUT}/${TAG}_bench_regional_n1.json"; cut -c1-200 "${OUT}/${TAG}_bench_train_under_rocprof.json"
head -8 "${OUT}/${TAG}_rocprofv3_kernel_stats_bench_train.csv" | cut -c1-200
