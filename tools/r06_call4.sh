#!/usr/bin/env bash
# r06 call 4: region_attn with 2-D query tiles + stride-2 convolutions on the raster kernel -- parity, kernel tables, same-box A/B
set -uo pipefail
OUT=gpurun_out/r06c4; mkdir -p $OUT
V=$PWD/mix-of-show_amd/_variants
python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "region_attention or stride2" 2>&1 | tail -6 > $OUT/tests_region_s2.txt
MOS_HIP_LIB=$V/libmos_hip_nsp2.so python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "region_attention" 2>&1 | tail -4 > $OUT/tests_region_nsp2.txt
for lib in base "" nsp2; do
  if [ -z "$lib" ]; then unset MOS_HIP_LIB; tag=tile2d; else export MOS_HIP_LIB=$V/libmos_hip_$lib.so; tag=$lib; fi
  python tools/bench_kernels.py --only region --iters 50 --ref 0 2>&1 | grep -i "region" > $OUT/kernels_region_$tag.txt
done
unset MOS_HIP_LIB
python tools/bench_kernels.py --only convs2 --iters 30 > $OUT/kernels_convs2.txt 2>&1
python tools/ab_switches.py --half regional "MOS_HIP_LIB=$V/libmos_hip_base.so MOS_CONV3X3_S2=0" "MOS_CONV3X3_S2=0" "" "MOS_HIP_LIB=$V/libmos_hip_nsp2.so" "MOS_HIP_LIB=$V/libmos_hip_base.so MOS_CONV3X3_S2=0" "MOS_CONV3X3_S2=0" "" "MOS_HIP_LIB=$V/libmos_hip_nsp2.so" > $OUT/ab_regional.txt 2>&1
python tools/ab_switches.py --half train "MOS_CONV3X3_S2=0" "" "MOS_CONV3X3_S2=0" "" > $OUT/ab_train.txt 2>&1
cat $OUT/tests_region_s2.txt $OUT/tests_region_nsp2.txt; for f in $OUT/kernels_*.txt; do echo "== $f"; cut -c1-200 $f; done; cat $OUT/ab_regional.txt $OUT/ab_train.txt
