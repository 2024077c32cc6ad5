#!/usr/bin/env bash
# Last call of round 4: HOST-side changes of the regional sampling path only (kernel sources unchanged since r04v2: the PMC passes,
# the training-side rocprofv3 table and the primitive parity of that call stay valid). Sampling-side GPU tests (all five parity
# cases -> parity_latents.json), same-box A/B against the tree one commit earlier (_variants/tree_before, same libmos_hip.so),
# the default bench line, rocprofv3 of the regional half.            bash tools/r04_final4.sh [tag]
set -u
TAG="${1:-r04v3}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
rm -f "$O/parity_latents.json"
echo "== 1. sampling-side end-to-end tests (parity cases, graph reuse / replay, written-out loop, smoke, one training parity test)"
timeout 420 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --durations=8 \
  -k "graft_smoke or regional or edlora_sd15 or pipeline_call_equals or training_steps_match" > "$O/${TAG}_gpu_tests_sampling.log" 2>&1
rc=$?; echo "rc=$rc"; grep -E "^\[parity\]|passed|failed|^FAILED|^E  " "$O/${TAG}_gpu_tests_sampling.log" | cut -c1-260 | tail -24
[ -f "$O/parity_latents.json" ] && cp "$O/parity_latents.json" profiles/parity_latents.json && \
  echo "profiles/parity_latents.json refreshed: $(python -c "import json; print(len(json.load(open('profiles/parity_latents.json'))['cases']), 'cases')")"
echo "== 2. same-box A/B, regional half: this tree vs the tree one commit earlier (interleaved)"
B="$ROOT/_variants/tree_before"
{
for rep in 1 2; do
  for side in after before; do
    d="$ROOT"; [ $side = before ] && d="$B"
    ( cd "$d" && timeout 200 python bench.py --mode regional --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$side', 'image %.1f ms' % d['value_ms_image'], 'latent %.1f ms' % d['value_ms_latent'], 'cold %.0f ms' % d['cold_call_ms'], 'reused', d['config'].get('graph_reused_across_calls'))" )
  done
done
} > "$O/${TAG}_ab_regional_host_side.txt" 2>&1
cat "$O/${TAG}_ab_regional_host_side.txt"
echo "== 3. default bench"
timeout 600 python bench.py --steps 20 --warmup 5 > "$O/${TAG}_bench_train_n1.json" 2> "$O/${TAG}_bench_train_n1.err"
tail -2 "$O/${TAG}_bench_train_n1.err"; cut -c1-260 "$O/${TAG}_bench_train_n1.json"
cd /tmp && export TMPDIR=/tmp
echo "== 4. rocprofv3 kernel stats: regional"
rm -rf /tmp/prof2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o b -- python "$ROOT/bench.py" --mode regional --steps 3 --warmup 1 --no-cpu-baseline \
    > "$O/${TAG}_bench_regional_under_rocprof.json" 2> "$O/${TAG}_rocprof_regional.err"
f=$(find /tmp/prof2 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -100 "$f" > "$O/${TAG}_rocprofv3_kernel_stats_bench_regional.csv"
cd "$ROOT"
grep -E "attn_fwd_kernelIDF16_Li40|MT16x16x128|CatArray|copyBuffer" "$O/${TAG}_rocprofv3_kernel_stats_bench_regional.csv" | cut -c1-150 | head -8
cut -c1-200 "$O/${TAG}_bench_regional_under_rocprof.json"
echo "== 5. fusion feature collection under no_grad with the batched time projections (if time is left)"
timeout 200 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -k "gradient_fusion_end_to_end or fusion_feature_collection" > "$O/${TAG}_gpu_tests_fusion.log" 2>&1
echo "rc=$?"; tail -2 "$O/${TAG}_gpu_tests_fusion.log"
