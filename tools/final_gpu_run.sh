#!/usr/bin/env bash
# One box: HBM-traffic PMC passes for the current kernel sources, parity of the DMA-ring conv / GEMM variants, kernel timings,
# whole-step A/B over the host-side switches (MOS_RING_MAX_WG, MOS_CONV3X3_MIN_PIXELS), then the default bench line and its
# rocprofv3 kernel stats under the best setting; copy gpurun_out/<tag>_* into profiles/ afterwards.   bash tools/final_gpu_run.sh <tag>
set -u
TAG="${1:-r02c}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== PMC traffic"
PMC_PASSES="fetch write" bash tools/pmc_collect.sh attn > "$O/${TAG}_pmc_run.log" 2>&1
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write > "$O/${TAG}_pmc_traffic.json" 2>> "$O/${TAG}_pmc_run.log"
if python -c "import json,sys; d=json.load(open('$O/${TAG}_pmc_traffic.json')); sys.exit(0 if d.get('kernels') else 1)"; then
  cp "$O/${TAG}_pmc_traffic.json" profiles/pmc_traffic.json; echo "pmc_traffic.json refreshed"
else echo "PMC FAILED"; tail -5 "$O/${TAG}_pmc_run.log"; fi
t() { timeout 400 python -m pytest tests/test_gpu_primitives.py -m gpu -q -x --timeout 300 -k "$1" 2>&1 | tail -2; }
echo "== parity (ring on)"; PAR=$(t "conv3x3 or lora_linear_fused"); echo "$PAR"
RING_OK=0; echo "$PAR" | grep -q " passed" && ! echo "$PAR" | grep -q "failed\|error" && RING_OK=1
echo "RING_OK=$RING_OK"
echo "== conv timings (ring on)"; timeout 300 python tools/bench_kernels.py --only conv --ref 0 --iters 12 2>/dev/null | grep -E "^B[0-9]"
echo "== gemm timings (ring on)"; timeout 300 python tools/bench_kernels.py --only gemm --ref 0 --iters 20 2>/dev/null | grep -E "^gemm_nt" | cut -c1-120
step() { timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-regional 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
BEST=""; BESTMS=999
try() {   # label, env assignments...
  local label="$1"; shift
  local ms; ms=$(env "$@" bash -c "$(declare -f step); step"); ms=${ms:-999}
  echo "== step [$label] $* -> $ms ms"
  if python -c "import sys; sys.exit(0 if float('$ms') < float('$BESTMS') else 1)"; then BESTMS=$ms; BEST="$*"; fi
}
try ring0 MOS_RING_MAX_WG=0 MOS_CONV3X3_MIN_PIXELS=4096
if [ "$RING_OK" = 1 ]; then
  try ring1 MOS_RING_MAX_WG=640 MOS_CONV3X3_MIN_PIXELS=4096
  try ring1-px1024 MOS_RING_MAX_WG=640 MOS_CONV3X3_MIN_PIXELS=1024
  try ring1-px256 MOS_RING_MAX_WG=640 MOS_CONV3X3_MIN_PIXELS=256
fi
echo "BEST: $BEST ($BESTMS ms)"; echo "$BEST" > "$O/${TAG}_best_env.txt"
echo "== default bench under [$BEST]"
env $BEST timeout 900 python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_train_n1.json" 2> "$O/${TAG}_bench_train_n1.err"
tail -3 "$O/${TAG}_bench_train_n1.err"; cut -c1-300 "$O/${TAG}_bench_train_n1.json"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
env $BEST timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o b -- python "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-regional \
    > "$O/${TAG}_bench_train_under_rocprof.json" 2> "$O/${TAG}_rocprof.err"
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -100 "$f" > "$O/${TAG}_rocprofv3_kernel_stats_bench_train.csv"
cd "$ROOT"
grep -E "attn_bwd_dkdv_kernelIDF16_Li40|conv3x3_nhwc_kernel" "$O/${TAG}_rocprofv3_kernel_stats_bench_train.csv" | cut -c1-160 | head -8
cut -c1-200 "$O/${TAG}_bench_train_under_rocprof.json"
