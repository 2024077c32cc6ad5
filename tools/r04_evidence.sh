#!/usr/bin/env bash
# Round-4 evidence run (VERDICT r03 "next round" item 1): the configs[3] fusion JSON line, the F2 parity check at sd15 /
# 14-concept scale, the JPEG-fed training step, the image-out regional latency with a real T2I-Adapter forward.      bash tools/r04_evidence.sh [tag]
set -u
TAG="${1:-r04}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
nproc > "$O/${TAG}_host_cores.txt"
echo "== new end-to-end tests (graph replay around eager steps, F2 at sd15 scale, sampling graph reuse)"
timeout 600 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --durations=6 \
  -k "replay_after_an_eager or one_sd15_level0 or graph_is_reused" > "$O/${TAG}_new_e2e_tests.log" 2>&1
echo "rc=$?"; grep -E "^\[parity\]|passed|failed|Error" "$O/${TAG}_new_e2e_tests.log" | cut -c1-400 | tail -12
echo "== configs[3]: gradient fusion of 14 synthetic ED-LoRAs (2 timed passes after 1 warm-up)"
python bench.py --mode fusion --concepts 14 --steps 2 --warmup 1 > "$O/${TAG}_bench_fusion.json" 2> "$O/${TAG}_bench_fusion.err"
echo "rc=$?"; tail -2 "$O/${TAG}_bench_fusion.err"; cut -c1-400 "$O/${TAG}_bench_fusion.json"
echo "== train step fed by the JPEG data pipeline (SURVEY 8(f).4), 8 and 16 workers"
for w in 8 16; do
  timeout 240 python bench.py --steps 20 --warmup 5 --data jpeg --workers $w --no-cpu-baseline --no-regional \
    > "$O/${TAG}_bench_train_jpeg_w${w}.json" 2> "$O/${TAG}_bench_train_jpeg_w${w}.err"
  echo "rc=$?"; grep "data pipeline ready" "$O/${TAG}_bench_train_jpeg_w${w}.err"; cut -c1-160 "$O/${TAG}_bench_train_jpeg_w${w}.json"
done
echo "== regional half: image out (adapter forward + VAE decode + PIL) and latent out"
timeout 400 python bench.py --mode regional --steps 3 --warmup 1 > "$O/${TAG}_bench_regional.json" 2> "$O/${TAG}_bench_regional.err"
echo "rc=$?"; tail -2 "$O/${TAG}_bench_regional.err"; cut -c1-330 "$O/${TAG}_bench_regional.json"
