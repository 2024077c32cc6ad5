#!/usr/bin/env bash
# Short GPU-box validation of a change set before the evidence run: the primitive tests named in $1 (a -k expression), the
# training-side end-to-end tests, one real-scale sampling parity test and a quick bench line.   bash tools/validate_gpu_run.sh '<-k expr>' [tag]
set -u
KEXPR="${1:-add_layernorm or groupnorm or layernorm}"
TAG="${2:-validate}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== primitives: $KEXPR"
timeout 300 python -m pytest tests/test_gpu_primitives.py -m gpu -x -q -s -k "$KEXPR" > "$O/${TAG}_primitives.log" 2>&1; echo "rc=$?"; tail -3 "$O/${TAG}_primitives.log"
grep -c "bit-identical.*True" "$O/${TAG}_primitives.log"; grep "bit-identical.*False" "$O/${TAG}_primitives.log" | head -5
echo "== end to end (training side, graph, smoke, one sampling parity test)"
timeout 600 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --durations=8 \
  -k "graft_smoke or training_steps_match or train_step_through_vae or hipgraph_step_equals or pipeline_call_equals or edlora_sd15_hot_path" \
  > "$O/${TAG}_e2e.log" 2>&1; echo "rc=$?"; grep -E "^\[parity\]|passed|failed|Error|error" "$O/${TAG}_e2e.log" | cut -c1-400 | tail -25
echo "== quick bench (train half, no CPU baseline)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-regional > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"; echo "rc=$?"
tail -3 "$O/${TAG}_bench.err"; cut -c1-300 "$O/${TAG}_bench.json"
