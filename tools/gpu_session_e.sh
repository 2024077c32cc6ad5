#!/usr/bin/env bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out; mkdir -p $O
echo "== primitives (lora/linear)"; timeout 600 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x -k "lora or linear or pack or softmax" 2>&1 | tail -3
echo "== gemm micro-benchmark (deep stages)"; timeout 300 python tools/bench_kernels.py --only gemm --iters 20 2>&1 | grep -v JSON | grep gemm_nt > $O/r02_kb_gemm3.txt; cat $O/r02_kb_gemm3.txt
echo "== gemm micro-benchmark (nodeep)"; MOS_HIP_LIB=$PWD/mix-of-show_amd/_variants/nodeep.so timeout 300 python tools/bench_kernels.py --only gemm --iters 20 2>&1 | grep -v JSON | grep "gemm_nt" | grep fused
echo "== true kernel durations (rocprofv3) of the gemm micro-benchmark"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kprof && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o k -- python $OLDPWD/tools/bench_kernels.py --only gemm --iters 20 > /dev/null 2>&1; f=$(find /tmp/kprof -name "*kernel_stats.csv" | head -1); cd $OLDPWD; [ -n "$f" ] && head -40 "$f" | cut -c1-200 | tee $O/r02_rocprofv3_kernel_stats_bench_kernels_gemm.csv
echo "== step A/B"; bash tools/try_variants.sh run_step 2>&1 | tail -20
