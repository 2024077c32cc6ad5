#!/usr/bin/env bash
# r06 call 8: what bounds region_attn -- additive ablation (variant builds: no compute / no staging / neither / loads without the LDS stores)
set -uo pipefail
OUT=gpurun_out/r06c8; mkdir -p $OUT
V=$PWD/mix-of-show_amd/_variants
for lib in "" ab1 ab2 ab3 ab4; do
  if [ -z "$lib" ]; then unset MOS_HIP_LIB; tag=full; else export MOS_HIP_LIB=$V/libmos_hip_$lib.so; tag=$lib; fi
  python tools/bench_kernels.py --only region --iters 100 --ref 0 2>&1 | grep "^region" | sed "s/^/$tag  /" >> $OUT/region_ablation.txt
done
cut -c1-130 $OUT/region_ablation.txt
