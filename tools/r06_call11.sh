#!/usr/bin/env bash
# r06 call 11: split-K convolution tile / workgroup-count variants on the low-resolution levels (same box)
set -uo pipefail
OUT=gpurun_out/r06c11; mkdir -p $OUT
V=$PWD/mix-of-show_amd/_variants
for lib in "" sp128x64w640 sp64x128w640 sp128x64w960 sp64x64w960; do
  if [ -z "$lib" ]; then unset MOS_HIP_LIB; tag=current_64x64_w640; else export MOS_HIP_LIB=$V/libmos_hip_$lib.so; tag=$lib; fi
  python tools/bench_kernels.py --only conv --iters 30 --ref 0 2>&1 | grep -E "^B[24] (1280|2560)->1280 (16x16|8x8|16x24|8x12)" | sed "s/^/$tag  /" >> $OUT/conv_splitk_variants.txt
done
unset MOS_HIP_LIB
cat $OUT/conv_splitk_variants.txt
