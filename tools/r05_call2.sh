#!/usr/bin/env bash
# Round 5, GPU call 2: the whole GPU suite on the tree with the halo convolution as default path, the materialised-probability
# controller kernels, the shipped 1024x2048 example, peaked-logit fixtures, single-rank RCCL; then the default bench line and
# the 1024x2048 regional line.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; cd "$ROOT"; O="$ROOT/gpurun_out"; mkdir -p "$O"
TAG="${1:-r05c2}"
rm -f "$O/parity_latents.json"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --durations=12 > "$O/${TAG}_gpu_tests.log" 2>&1
echo "pytest rc=$?"; tail -40 "$O/${TAG}_gpu_tests.log" | cut -c1-220
grep -E "^\[parity\].*(peaked|1024x2048|full-map|controller)" "$O/${TAG}_gpu_tests.log" | cut -c1-400
[ -f "$O/parity_latents.json" ] && cp "$O/parity_latents.json" profiles/parity_latents.json
echo "== default bench"
timeout 600 python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_train_n1.json" 2> "$O/${TAG}_bench_train_n1.err"
echo "rc=$?"; tail -2 "$O/${TAG}_bench_train_n1.err"; python - "$O/${TAG}_bench_train_n1.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'], d['whole_step'])
print([(k['kernel'], k['ms'], k['frac_of_mfma_peak']) for k in d['dominant_kernels_by_name'][:8]])
print({k: d[k] for k in d if k.startswith('regional_')})
print([(k['kernel'], k['ms']) for k in d['regional']['dominant_kernels_by_name'][:8]])
PY
echo "== shipped regional example 1024x2048"
timeout 600 python bench.py --mode regional --height 1024 --width 2048 --steps 2 --warmup 1 > "$O/${TAG}_bench_regional_1024x2048.json" 2> "$O/${TAG}_bench_regional_1024x2048.err"
echo "rc=$?"; tail -3 "$O/${TAG}_bench_regional_1024x2048.err"; python - "$O/${TAG}_bench_regional_1024x2048.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['value_ms_latent'], d['cold_call_ms'], d['roofline'], d['attention_path'])
print([(k['kernel'], k['ms']) for k in d['dominant_kernels_by_name'][:8]])
PY
