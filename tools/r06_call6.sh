#!/usr/bin/env bash
# r06 call 6: GroupNorm statistics from the convolution epilogue -- parity + same-box A/B (MOS_GN_FROM_CONV)
set -uo pipefail
OUT=gpurun_out/r06c6; mkdir -p $OUT
python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "leaves_groupnorm or conv3x3_nhwc or stride2 or groupnorm" 2>&1 | tail -8 > $OUT/tests_conv_gn.txt
python -m pytest tests/test_gpu_end_to_end.py -m gpu -x -q -k "train_step or smoke or teacher or regional" 2>&1 | tail -6 > $OUT/tests_e2e.txt
python tools/ab_switches.py --half train "MOS_GN_FROM_CONV=0" "" "MOS_GN_FROM_CONV=0" "" --kernels conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,groupnorm_finalize,groupnorm_finalize_pre,groupnorm_fused > $OUT/ab_train.txt 2>&1
python tools/ab_switches.py --half regional "MOS_GN_FROM_CONV=0" "" "MOS_GN_FROM_CONV=0" "" --kernels conv3x3,gemm_nt,groupnorm_apply,groupnorm_stats,groupnorm_finalize,groupnorm_finalize_pre,groupnorm_fused > $OUT/ab_regional.txt 2>&1
cat $OUT/tests_conv_gn.txt $OUT/tests_e2e.txt; cut -c1-230 $OUT/ab_train.txt $OUT/ab_regional.txt
