#!/usr/bin/env bash
# r06 call 7: where the replayed training step's device time goes (kernel trace: busy vs idle gaps) + A/B of GN statistics on every map
set -uo pipefail
OUT=gpurun_out/r06c7; mkdir -p $OUT
python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "leaves_groupnorm" 2>&1 | tail -3 > $OUT/tests_gn.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06c7_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-regional > $GRAFT_REPO_ROOT/$OUT/trace_bench.json 2> $GRAFT_REPO_ROOT/$OUT/trace_bench.err
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py gpurun_out/r06c7_trace > $OUT/trace_gaps_train.txt 2>&1
rm -rf gpurun_out/r06c7_trace
python tools/ab_switches.py --half regional "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" --kernels conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,groupnorm_finalize,groupnorm_finalize_pre,groupnorm_fused > $OUT/ab_regional_gn_always.txt 2>&1
python tools/ab_switches.py --half train "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" --kernels conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,groupnorm_finalize_pre,groupnorm_fused,groupnorm_bwd_fused > $OUT/ab_train_gn_always.txt 2>&1
cat $OUT/tests_gn.txt; cut -c1-200 $OUT/trace_gaps_train.txt; cut -c1-230 $OUT/ab_regional_gn_always.txt $OUT/ab_train_gn_always.txt
