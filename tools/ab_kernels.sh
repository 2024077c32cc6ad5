#!/usr/bin/env bash
# Same-box A/B of library variants on the kernel micro-benchmark:
#   bash tools/ab_kernels.sh <tag> "<bench_kernels --only list>" name=path.so [name=path.so ...]
# Each variant is loaded through MOS_HIP_LIB (mixofshow/hip/lib.py); one table per variant in gpurun_out/<tag>_ab_<name>.txt
set -u
TAG="$1"; ONLY="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
O="$ROOT/gpurun_out"; mkdir -p "$O"
for round in 1 2; do
  for kv in "$@"; do
    name="${kv%%=*}"; lib="${kv#*=}"
    for what in $ONLY; do
      MOS_HIP_LIB="$ROOT/$lib" timeout 300 python "$ROOT/tools/bench_kernels.py" --only "$what" --iters 12 --ref 0 2>&1 \
        | grep -E "^(attn|region|gemm|lora|conv|gn_|gram|lsq)" | cut -c1-130 | sed "s/^/[$name r$round] /" >> "$O/${TAG}_ab_${name}.txt"
    done
  done
done
for kv in "$@"; do name="${kv%%=*}"; echo "== $name"; cat "$O/${TAG}_ab_${name}.txt"; done
