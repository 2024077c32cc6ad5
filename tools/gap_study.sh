#!/usr/bin/env bash
# the driver's fresh box ran the step 15 % slower than the builder's boxes. This script is the driver's
# command as the FIRST process of a fresh lease, three times back to back, then a long steady-state run: every compact line,
# every stderr log (device-side per-step spread, clocks during the timed region) kept under gpurun_out/<tag>/.
set -uo pipefail
TAG="${1:-gap}"
OUT="gpurun_out/${TAG}"
mkdir -p "${OUT}"
(rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>&1 | head -60) > "${OUT}/rocm_smi_before.txt" || true
for i in 1 2 3; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > "${OUT}/run${i}.json" 2> "${OUT}/run${i}.err"
  echo "run ${i}: rc $? bytes $(wc -c < "${OUT}/run${i}.json")"
  cp gpurun_out/bench_full.json "${OUT}/run${i}_full.json" 2>/dev/null || true
  grep -E "timed region|captured|cold|regional" "${OUT}/run${i}.err" | head -8
done
python3 bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline --no-regional > "${OUT}/long.json" 2> "${OUT}/long.err"
grep -E "timed region" "${OUT}/long.err"
(rocm-smi --showclocks --showpower 2>&1 | head -40) > "${OUT}/rocm_smi_after.txt" || true
tail -c 400 "${OUT}/run1.json"
