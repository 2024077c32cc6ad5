#!/usr/bin/env bash
# One GPU-box pass that regenerates the bench evidence under profiles/ (copy gpurun_out/<tag>_* there afterwards):
#   bash tools/final_gpu_run.sh <tag>
# 1. the default bench line (both metric halves + cpu baselines), 2. rocprofv3 kernel stats of the same command,
# 3. PMC passes (FETCH_SIZE / WRITE_SIZE, SQ instruction mix) over the attention micro-benchmark -> pmc_traffic.json
set -u
TAG="${1:-r02}"
ROOT="$(pwd)"
OUT="${ROOT}/gpurun_out"
mkdir -p "${OUT}"
timeout 900 python bench.py --steps 20 --warmup 5 > "${OUT}/${TAG}_bench_train_n1.json" 2> "${OUT}/${TAG}_bench_train_n1.err"
tail -4 "${OUT}/${TAG}_bench_train_n1.err"; cut -c1-400 "${OUT}/${TAG}_bench_train_n1.json"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python "${ROOT}/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-regional \
    > "${OUT}/${TAG}_bench_train_under_rocprof.json" 2> "${OUT}/${TAG}_rocprof.err"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -100 "$f" > "${OUT}/${TAG}_rocprofv3_kernel_stats_bench_train.csv"
cd "${ROOT}"
grep -E "attn_bwd_dkdv_kernelIDF16_Li40|attn_bwd_dq_kernelIDF16_Li40|attn_fwd_kernelIDF16_Li40" "${OUT}/${TAG}_rocprofv3_kernel_stats_bench_train.csv" | cut -c1-160
cut -c1-200 "${OUT}/${TAG}_bench_train_under_rocprof.json"
bash tools/pmc_collect.sh attn > "${OUT}/${TAG}_pmc_run.log" 2>&1
cp "${OUT}/pmc_attn.txt" "${OUT}/${TAG}_pmc_attention_kernels.txt" 2>/dev/null
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write > "${OUT}/${TAG}_pmc_traffic.json" 2>> "${OUT}/${TAG}_pmc_run.log"; cat "${OUT}/${TAG}_pmc_traffic.json"
grep -A24 "attn_bwd_dkdv_kernel f16 40 grid=262144" "${OUT}/${TAG}_pmc_attention_kernels.txt" | head -30
