#!/usr/bin/env bash
# Build a tuning variant of libmos_hip.so: [VARIANT_SRC=mos_gemm] bash tools/build_variant.sh <name> <extra hipcc flags...>
# -> mix-of-show_amd/_variants/<name>.so ; run with MOS_HIP_LIB=<that path>
set -euo pipefail
NAME="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; C="${ROOT}/mix-of-show_amd/csrc"; V="${ROOT}/mix-of-show_amd/_variants"
mkdir -p "$V" "$C/_build"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1"
# VARIANT_SRC selects the translation unit the extra flags apply to (default mos_attn); the others come from _build/
SRC="${VARIANT_SRC:-mos_attn}"
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$C/${SRC}.hip" -o "$V/${NAME}_${SRC}.o"
OBJS=""
for f in mos_api mos_gemm mos_attn mos_gram mos_norm mos_elem mos_conv; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS $V/${NAME}_${SRC}.o"; else OBJS="$OBJS $C/_build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/${NAME}.so" $OBJS
rm -f "$V/${NAME}_${SRC}.o"; echo "built $V/${NAME}.so"
