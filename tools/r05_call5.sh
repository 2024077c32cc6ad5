#!/usr/bin/env bash
# Round 5, GPU call 5: do torch's TunableOp choices for the library GEMMs that stay on hipBLASLt (feed-forward layers, time
# projections) move the step? default / tuning pass (writes the CSV) / tuned replay, train and regional halves.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG=r05c5
CSV="$O/${TAG}_tunableop_results.csv"
run() { # label, env...
  local label="$1"; shift
  env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>"$O/${TAG}_${label}.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$label] train', d['value'], 'img/s', d['ms_per_step'], 'ms; regional image', d['regional_ms_image'], 'latent', d['regional_ms_latent'], 'cold', d['regional_cold_call_ms'])"
}
{
run default A=1
run tuning PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME="$CSV" PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
ls -la "$O"/${TAG}_tunableop_results*.csv 2>/dev/null | head; F=$(ls "$O"/${TAG}_tunableop_results*.csv 2>/dev/null | head -1)
[ -n "$F" ] && { wc -l "$F"; head -12 "$F" | cut -c1-200; }
run tuned PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME="$CSV"
run default2 A=1
run tuned2 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=0 PYTORCH_TUNABLEOP_FILENAME="$CSV"
} 2>&1 | tee "$O/${TAG}_tunableop_ab.txt"
tail -3 "$O/${TAG}_tuning.err"
