#!/usr/bin/env bash
# r06 call 5: FF / CLIP-MLP GEMMs hipBLASLt vs the library kernel (forward and backward-data); region_attn same-box A/B (base vs 2-D tiles vs NSP=2)
set -uo pipefail
OUT=gpurun_out/r06c5; mkdir -p $OUT
V=$PWD/mix-of-show_amd/_variants
python tools/bench_kernels.py --only ffgemm --iters 30 2>&1 | grep -v amdgpu.ids > $OUT/kernels_ffgemm.txt
for lib in base "" nsp2; do
  if [ -z "$lib" ]; then unset MOS_HIP_LIB; tag=tile2d; else export MOS_HIP_LIB=$V/libmos_hip_$lib.so; tag=$lib; fi
  python tools/bench_kernels.py --only region --iters 50 --ref 0 2>&1 | grep "^region" > $OUT/kernels_region_$tag.txt
done
unset MOS_HIP_LIB
MOS_HIP_LIB=$V/libmos_hip_nsp2.so python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "region_attention" 2>&1 | tail -3 > $OUT/tests_region_nsp2.txt
python tools/ab_switches.py --half regional "MOS_HIP_LIB=$V/libmos_hip_base.so" "" "MOS_HIP_LIB=$V/libmos_hip_nsp2.so" "MOS_HIP_LIB=$V/libmos_hip_base.so" "" "MOS_HIP_LIB=$V/libmos_hip_nsp2.so" > $OUT/ab_regional_region_attn.txt 2>&1
cut -c1-160 $OUT/kernels_ffgemm.txt; for f in $OUT/kernels_region_*.txt; do echo "== $f"; cut -c1-120 $f; done; cat $OUT/tests_region_nsp2.txt; cut -c1-220 $OUT/ab_regional_region_attn.txt
