#!/usr/bin/env bash
# Build tuning variants of the kernels HERE (no GPU needed), then evaluate them in ONE gpurun call:
#   bash tools/try_variants.sh build                  # -> mix-of-show_amd/_variants/<name>.so (travel with the snapshot)
#   gpurun -- 'bash tools/try_variants.sh run'        # attention variants: parity + kernel micro-benchmark per variant
#   gpurun -- 'bash tools/try_variants.sh run_step'   # step-level A/B (bench.py, same box): base vs each "step" variant
# A micro-benchmark win does not always survive in the full step (the fused LoRA-gradient reduction was -23 % in the
# micro-benchmark and -3.4 % images/s in the step): always confirm with run_step before changing a default.
# Entries: "name|translation unit|extra hipcc flags|attn or step"
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
VARIANTS=(
  "occ1|mos_attn|-DMOS_DQ_OCC=1|attn"                        # dQ d=40 at one block per CU
  "deep768|mos_gemm|-DMOS_GEMM_DEEP_MAX_WG=768|step"         # deep-stage GEMM variants for all grids <= 768 workgroups (measured slower)
  "g512|mos_gemm|-DMOS_GRAD_TARGET_WG=512|step"              # fused LoRA-gradient kernel: fewer, longer token chunks
  "g2048|mos_gemm|-DMOS_GRAD_TARGET_WG=2048|step"            # ... more, shorter chunks
)
SHAPES="Nq4096 Nkv4096\|d80 B4 H8 Nq1024 Nkv1024\|B2 H8"
case "${1:-}" in
  build)
    bash "${ROOT}/mix-of-show_amd/csrc/build.sh" >/dev/null   # fresh objects of the other translation units
    for v in "${VARIANTS[@]}"; do
      IFS='|' read -r name src flags _ <<< "$v"
      # shellcheck disable=SC2086
      VARIANT_SRC="$src" bash "${ROOT}/tools/build_variant.sh" "$name" $flags 2>&1 | tail -1
    done ;;
  run)
    cd "${ROOT}"
    echo "== base"; python tools/bench_kernels.py --only attn --iters 10 2>&1 | grep -v JSON | grep "$SHAPES"
    for v in "${VARIANTS[@]}"; do
      IFS='|' read -r name _ _ kind <<< "$v"
      [ "$kind" = attn ] || continue
      so="${ROOT}/mix-of-show_amd/_variants/${name}.so"
      [ -f "$so" ] || { echo "== $name: not built"; continue; }
      echo "== $name"
      MOS_HIP_LIB="$so" timeout 120 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "attention" 2>&1 | tail -2
      MOS_HIP_LIB="$so" timeout 60 python tools/bench_kernels.py --only attn --iters 10 2>&1 | grep -v JSON | grep "$SHAPES"
    done ;;
  run_step)
    cd "${ROOT}"
    step() { timeout 250 python bench.py --no-cpu-baseline --steps 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], 'images/s', d['ms_per_step'], 'ms')"; }
    echo "== base"; step; step
    for v in "${VARIANTS[@]}"; do
      IFS='|' read -r name _ _ kind <<< "$v"
      [ "$kind" = step ] || [ "${2:-}" = all ] || continue
      so="${ROOT}/mix-of-show_amd/_variants/${name}.so"
      [ -f "$so" ] || { echo "== $name: not built"; continue; }
      echo "== $name"
      MOS_HIP_LIB="$so" timeout 120 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -k "lora or linear" 2>&1 | tail -1
      export MOS_HIP_LIB="$so"; step; unset MOS_HIP_LIB
    done ;;
  *) echo "usage: $0 build | run | run_step [all]"; exit 2 ;;
esac
