#!/usr/bin/env bash
# First GPU call of round 5: what round 4 left unmeasured. (1) the whole GPU suite on the final round-4 tree (the L-BFGS history
# form and the per-layer loads of gradient_fusion.py were validated on the CPU only), (2) configs[3] with the per-stage breakdown
# (`stage_seconds_last_pass`, `solve_seconds_last_pass`) and a same-box A/B of the layer solves against the tree before those
# changes -- build it first:   mkdir -p _variants/tree_before && git archive 6ad1be8 | tar -x -C _variants/tree_before &&
#                              cp mix-of-show_amd/libmos_hip.so _variants/tree_before/mix-of-show_amd/
# (3) the default bench line + rocprofv3 of both halves.                              bash tools/r05_first_call.sh [tag]
set -u
TAG="${1:-r05a}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
rm -f "$O/parity_latents.json"
echo "== 1. full GPU test suite"
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 > "$O/${TAG}_gpu_tests.log" 2>&1
echo "rc=$?"; tail -12 "$O/${TAG}_gpu_tests.log" | cut -c1-200
[ -f "$O/parity_latents.json" ] && cp "$O/parity_latents.json" profiles/parity_latents.json
echo "== 2. configs[3]: three passes (cold, warm, profiled) with stage / solve seconds"
timeout 300 python bench.py --mode fusion --concepts 14 --steps 2 --warmup 1 --no-cpu-baseline > "$O/${TAG}_bench_fusion.json" 2> "$O/${TAG}_bench_fusion.err"
echo "rc=$?"; grep "fusion pass" "$O/${TAG}_bench_fusion.err"
python - "$O/${TAG}_bench_fusion.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d.get('stage_seconds_last_pass'), d.get('solve_seconds_last_pass'))
PY
B="$ROOT/_variants/tree_before"
if [ -d "$B" ]; then
  echo "== same box, one cold + profiled pass each: this tree / the tree before the round-4 fusion host changes"
  for side in after before; do
    d="$ROOT"; [ $side = before ] && d="$B"
    ( cd "$d" && timeout 120 python bench.py --mode fusion --concepts 14 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$side', d['value'], d.get('solve_seconds_last_pass'), d.get('stage_seconds_last_pass'))" )
  done > "$O/${TAG}_ab_fusion_host_side.txt" 2>&1
  cat "$O/${TAG}_ab_fusion_host_side.txt"
fi
echo "== 3. default bench"
timeout 600 python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_train_n1.json" 2> "$O/${TAG}_bench_train_n1.err"
tail -2 "$O/${TAG}_bench_train_n1.err"; cut -c1-260 "$O/${TAG}_bench_train_n1.json"
cd /tmp && export TMPDIR=/tmp
for half in train regional; do
  echo "== rocprofv3 kernel stats: $half"
  rm -rf /tmp/prof_$half
  extra="--steps 6 --warmup 2 --no-cpu-baseline --no-regional"; [ $half = regional ] && extra="--mode regional --steps 3 --warmup 1 --no-cpu-baseline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$half -o b -- python "$ROOT/bench.py" $extra \
      > "$O/${TAG}_bench_${half}_under_rocprof.json" 2> "$O/${TAG}_rocprof_${half}.err"
  f=$(find /tmp/prof_$half -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$O/${TAG}_rocprofv3_kernel_stats_bench_${half}.csv"
  # the launch ORDER of the last steps (which kernels surround the ~76 strided copies per training step that the CPU replay of
  # the host code does not show? DESIGN 8): name + start + duration of the last 9000 launches
  t=$(find /tmp/prof_$half -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python - "$t" "$O/${TAG}_kernel_order_${half}.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
with open(sys.argv[2], 'w') as f:
    for r in rows[-9000:]:
        f.write(f"{r['Kernel_Name'][:110]},{r['Start_Timestamp']},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])}\n")
PY
done
cd "$ROOT"
