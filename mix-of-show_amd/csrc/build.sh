#!/usr/bin/env bash
# Build libmos_hip.so for gfx950 (MI355X). Cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libmos_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast"
mkdir -p "${HERE}/_build"
pids=()
for f in mos_api mos_gemm mos_attn mos_gram; do
  ( ${HIPCC} ${FLAGS} -c "${HERE}/${f}.hip" -o "${HERE}/_build/${f}.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
${HIPCC} --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${HERE}"/_build/mos_api.o "${HERE}"/_build/mos_gemm.o "${HERE}"/_build/mos_attn.o "${HERE}"/_build/mos_gram.o
echo "built ${OUT}"
