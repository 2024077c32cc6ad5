#!/usr/bin/env bash
# r06 call 10: the one-launch GroupNorm from the producer's statistics -- parity, then same-box A/B: statistics on the large maps only (1) vs on every map (2)
set -uo pipefail
OUT=gpurun_out/r06c10; mkdir -p $OUT
python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "leaves_groupnorm or groupnorm" 2>&1 | tail -4 > $OUT/tests_gn.txt
python tools/ab_switches.py --half regional "MOS_GN_FROM_CONV=0" "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" "MOS_GN_FROM_CONV=0" "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" --kernels conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,groupnorm_finalize,groupnorm_pre,groupnorm_fused > $OUT/ab_regional_gn_pre.txt 2>&1
python tools/ab_switches.py --half train "MOS_GN_FROM_CONV=0" "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" "MOS_GN_FROM_CONV=0" "MOS_GN_FROM_CONV=1" "MOS_GN_FROM_CONV=2" --kernels conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,groupnorm_pre,groupnorm_fused,groupnorm_bwd_fused > $OUT/ab_train_gn_pre.txt 2>&1
cat $OUT/tests_gn.txt; cut -c1-230 $OUT/ab_regional_gn_pre.txt $OUT/ab_train_gn_pre.txt
