#!/usr/bin/env bash
# Build a tuning variant of libmos_hip.so: bash tools/build_variant.sh <name> <extra hipcc flags...>
# -> mix-of-show_amd/_variants/<name>.so ; run with MOS_HIP_LIB=<that path>
set -euo pipefail
NAME="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; C="${ROOT}/mix-of-show_amd/csrc"; V="${ROOT}/mix-of-show_amd/_variants"
mkdir -p "$V" "$C/_build"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$C/mos_attn.hip" -o "$V/${NAME}_attn.o"
OBJS="$V/${NAME}_attn.o"
for f in mos_api mos_gemm mos_gram mos_norm; do OBJS="$OBJS $C/_build/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$V/${NAME}.so" $OBJS
rm -f "$V/${NAME}_attn.o"; echo "built $V/${NAME}.so"
