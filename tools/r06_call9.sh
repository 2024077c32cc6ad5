#!/usr/bin/env bash
# r06 call 9: region_attn occupancy variants (d40: 2 resident sources, 3 workgroups per CU = the whole grid in ONE round; d80: 2 per CU)
set -uo pipefail
OUT=gpurun_out/r06c9; mkdir -p $OUT
V=$PWD/mix-of-show_amd/_variants
for lib in "" occa occc; do
  if [ -z "$lib" ]; then unset MOS_HIP_LIB; tag=current; else export MOS_HIP_LIB=$V/libmos_hip_$lib.so; tag=$lib; fi
  python tools/bench_kernels.py --only region --iters 100 --ref 0 2>&1 | grep "^region" | sed "s/^/$tag  /" >> $OUT/region_occupancy.txt
done
MOS_HIP_LIB=$V/libmos_hip_occc.so python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "region_attention" 2>&1 | tail -3 >> $OUT/region_occupancy.txt
unset MOS_HIP_LIB
python tools/ab_switches.py --half regional "" "MOS_HIP_LIB=$V/libmos_hip_occc.so" "" "MOS_HIP_LIB=$V/libmos_hip_occc.so" --kernels conv3x3,gemm_nt,attn_fwd,region_attn >> $OUT/region_occupancy.txt 2>&1
cut -c1-170 $OUT/region_occupancy.txt
