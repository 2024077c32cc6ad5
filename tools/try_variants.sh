#!/usr/bin/env bash
# Build tuning variants of the attention kernels HERE (no GPU needed), then evaluate them in ONE gpurun call:
#   bash tools/try_variants.sh build            # -> mix-of-show_amd/_variants/{v3,v2nk2,...}.so (travel with the snapshot)
#   gpurun -- 'bash tools/try_variants.sh run'  # per variant: attention/region parity, then the kernel micro-benchmark
# Variants are listed in VARIANTS below as "name|extra hipcc flags for mos_attn.hip".
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
VARIANTS=(
  "v3|-DMOS_DKDV_V2=3"                       # three-stage NK=1 pipelined dK/dV (mos_attn_dkdv_v2.inc)
  "v2nk1|-DMOS_DKDV_V2=1 -DMOS_DKDV_V2_NK=1" # two-stage pipelined dK/dV, one key group
  "v2nk2|-DMOS_DKDV_V2=1 -DMOS_DKDV_V2_NK=2" # two-stage, two key groups (spills today)
  "dq8|-DMOS_DQ_NW=8"                        # 8-wave dQ blocks
  "fold|-DMOS_DKDV_FOLD=1"                   # dK/dV d=40: -lse/scale and -D folded into the pad columns of the MFMA contraction
  "lsum|-DMOS_FWD_LSUM=1"                    # forward d<=80: row sums from the P.V MFMA (ones row in V^T padding), -14 % main-loop VALU
  "noslp|-fno-slp-vectorize"                 # no v_pk_{mul,add}_f32 (2804 -> 24 in mos_attn; the guide calls packed f32 VALU an anti-lever beside MFMAs)
)
case "${1:-}" in
  build)
    bash "${ROOT}/mix-of-show_amd/csrc/build.sh" >/dev/null   # fresh objects of the other translation units
    for v in "${VARIANTS[@]}"; do
      name="${v%%|*}"; flags="${v#*|}"
      # shellcheck disable=SC2086
      bash "${ROOT}/tools/build_variant.sh" "$name" $flags 2>&1 | tail -1
    done ;;
  run)
    cd "${ROOT}"
    echo "== base"; python tools/bench_kernels.py --only attn --iters 10 2>&1 | grep -v JSON | grep "Nq4096 Nkv4096\|d80 B4 H8 Nq1024 Nkv1024\|B2 H8"
    for v in "${VARIANTS[@]}"; do
      name="${v%%|*}"; so="${ROOT}/mix-of-show_amd/_variants/${name}.so"
      [ -f "$so" ] || { echo "== $name: not built"; continue; }
      echo "== $name"
      MOS_HIP_LIB="$so" timeout 120 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "attention" 2>&1 | tail -2
      MOS_HIP_LIB="$so" timeout 60 python tools/bench_kernels.py --only attn --iters 10 2>&1 | grep -v JSON | grep "Nq4096 Nkv4096\|d80 B4 H8 Nq1024 Nkv1024\|B2 H8"
    done ;;
  *) echo "usage: $0 build|run"; exit 2 ;;
esac
