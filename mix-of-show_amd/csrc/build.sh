#!/usr/bin/env bash
# Build libmos_hip.so for gfx950 (MI355X). Cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# MOS_OUT / MOS_BUILD_DIR / MOS_ATTN_SRC / MOS_ATTN_FLAGS / MOS_CONV_FLAGS: kernel-variant builds for same-box A/B runs (load with MOS_HIP_LIB=...)
OUT="${MOS_OUT:-${HERE}/../libmos_hip.so}"
BUILD="${MOS_BUILD_DIR:-${HERE}/_build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast"
SRCS="mos_api mos_gemm mos_attn mos_probs mos_gram mos_norm mos_elem mos_conv"
mkdir -p "${BUILD}"
pids=()
# mos_attn only: at one wave per SIMD hipcc selects the AccVGPR form of every MFMA and then copies the accumulators the
# VALU touches (softmax) through v_accvgpr_read/write -- 6..10 copies per MFMA in the d = 80/160 and region kernels.
# The VGPR form keeps them in arch VGPRs (main-loop VALU count -40 %; region kernels -15..-26 % measured, parity
# unchanged). Two-wave kernels (d = 40) compile to the same code either way.
EXTRA_mos_attn="-mllvm -amdgpu-mfma-vgpr-form=1 ${MOS_ATTN_FLAGS:-}"
# mos_gemm / mos_conv (round 4): in the AccVGPR form the DMA-ring variants (3-4 stage K loops of the small grids) come out of
# register allocation with 20-50 v_accvgpr_read/write/mov per K tile rotating the accumulator tuples (64x64 GEMM ring: 42 VALU
# for 8 MFMAs, fused 64x128 ring: 94 for 20); the VGPR form has none (22 / 38). VALU and MFMA time add up on this chip.
# MOS_MFMA_FORM_FLAGS="" restores the AccVGPR form for an A/B build.
EXTRA_mos_gemm="${MOS_MFMA_FORM_FLAGS--mllvm -amdgpu-mfma-vgpr-form=1} ${MOS_GEMM_FLAGS:-}"
# MOS_CONV_FLAGS="-DMOS_CONV_NO_HALO": a variant build whose 3x3 convolutions all take the raster form of rounds 2-4 (same-box A/B
# of the halo-staged form; load with MOS_HIP_LIB=...). Build-time only: the library reads no environment variable.
EXTRA_mos_conv="${MOS_MFMA_FORM_FLAGS--mllvm -amdgpu-mfma-vgpr-form=1} ${MOS_CONV_FLAGS:-}"
for f in ${SRCS}; do
  extra_var="EXTRA_${f}"
  src="${HERE}/${f}.hip"
  if [ "${f}" = mos_attn ] && [ -n "${MOS_ATTN_SRC:-}" ]; then src="${MOS_ATTN_SRC}"; fi
  ( ${HIPCC} ${FLAGS} ${!extra_var:-} -I"${HERE}" -c "${src}" -o "${BUILD}/${f}.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
OBJS=""
for f in ${SRCS}; do OBJS="${OBJS} ${BUILD}/${f}.o"; done
${HIPCC} --offload-arch=gfx950 -shared -fPIC -o "${OUT}" ${OBJS}
echo "built ${OUT}"
