#!/usr/bin/env bash
# rocprofv3 kernel stats of the bench command (GPU box): bash tools/rocprof_bench.sh <tag> [bench args...]
set -u
TAG="${1:-r01}"; shift || true
ROOT="$(pwd)"; OUT="${ROOT}/gpurun_out"; mkdir -p "${OUT}"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python "${ROOT}/bench.py" --steps 3 --warmup 2 --no-cpu-baseline "$@" \
    > "${OUT}/${TAG}_bench_train_under_rocprof.json" 2> "${OUT}/${TAG}_rocprof.err"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -120 "$f" > "${OUT}/${TAG}_rocprofv3_kernel_stats_bench_train.csv"
ls -la /tmp/prof/* | head; cut -c1-200 "${OUT}/${TAG}_bench_train_under_rocprof.json"
head -40 "${OUT}/${TAG}_rocprofv3_kernel_stats_bench_train.csv" | cut -c1-170
