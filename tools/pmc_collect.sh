#!/usr/bin/env bash
# PMC passes over the kernel micro-benchmark (rocprofv3, counters only + kernel trace, one pass per counter group —
# see /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"). Writes a per-kernel summary to gpurun_out/.
#   bash tools/pmc_collect.sh attn   (or gemm / region / gram / conv; PMC_BENCH_ARGS="--ref 0" skips the library references)
set -u
WHAT="${1:-attn}"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT"
cd /tmp
pass() {   # name, counters...
  local name="$1"; shift
  rm -rf "/tmp/pmc_$name"
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "/tmp/pmc_$name" -o p -- \
      python "$ROOT/tools/bench_kernels.py" --only "$WHAT" --iters 3 ${PMC_BENCH_ARGS:-} > "/tmp/pmc_$name.log" 2>&1 || tail -5 "/tmp/pmc_$name.log"
}
# PMC_PASSES="fetch write" limits the run to the HBM-traffic counters (tools/pmc_traffic.py needs only those)
PASSES="${PMC_PASSES:-sq1 sq2 fetch write}"
for p in $PASSES; do
  case "$p" in
    sq1) pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS ;;
    sq2) pass sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES ;;
    fetch) pass fetch FETCH_SIZE ;;
    write) pass write WRITE_SIZE ;;
  esac
done
if [ "$PASSES" != "fetch write" ]; then
  python "$ROOT/tools/pmc_summary.py" $(for p in $PASSES; do echo /tmp/pmc_$p; done) > "$OUT/pmc_${WHAT}.txt" 2>&1
  tail -60 "$OUT/pmc_${WHAT}.txt"
fi
