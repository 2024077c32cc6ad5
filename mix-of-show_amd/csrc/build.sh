#!/usr/bin/env bash
# Build libmos_hip.so for gfx950 (MI355X). Cross-compiles without a GPU.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libmos_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast"
SRCS="mos_api mos_gemm mos_attn mos_gram mos_norm"
mkdir -p "${HERE}/_build"
pids=()
for f in ${SRCS}; do
  ( ${HIPCC} ${FLAGS} -c "${HERE}/${f}.hip" -o "${HERE}/_build/${f}.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
OBJS=""
for f in ${SRCS}; do OBJS="${OBJS} ${HERE}/_build/${f}.o"; done
${HIPCC} --offload-arch=gfx950 -shared -fPIC -o "${OUT}" ${OBJS}
echo "built ${OUT}"
