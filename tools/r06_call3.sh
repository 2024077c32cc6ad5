#!/usr/bin/env bash
# r06 call 3: GPU checks of the round's host-side changes + same-box A/B of the out-projection residual epilogue
set -uo pipefail
OUT=gpurun_out/r06c3; mkdir -p $OUT
python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "region_attention or materialised" 2>&1 | tail -4 > $OUT/tests_primitives.txt
python -m pytest tests/test_gpu_end_to_end.py -m gpu -x -q -k "controller or regional or train_step or smoke or hipgraph or graph" 2>&1 | tail -6 > $OUT/tests_e2e.txt
python tools/ab_switches.py --half train "MOS_ATTN_OUT_RESIDUAL=0" "" "MOS_ATTN_OUT_RESIDUAL=0" "" > $OUT/ab_train.txt 2>&1
python tools/ab_switches.py --half regional "MOS_ATTN_OUT_RESIDUAL=0" "" "MOS_ATTN_OUT_RESIDUAL=0" "" > $OUT/ab_regional.txt 2>&1
tail -5 $OUT/tests_primitives.txt $OUT/tests_e2e.txt; cat $OUT/ab_train.txt $OUT/ab_regional.txt
