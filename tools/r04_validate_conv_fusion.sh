#!/usr/bin/env bash
# Round-4, third GPU call: the 128-wide split-K convolution plan, the fp64-MFMA least-squares kernel, per-concept Gram batches,
# the new defaults (conv threshold 0, batched time projections, FF2 rule).      bash tools/r04_validate_conv_fusion.sh [tag]
set -u
TAG="${1:-r04c}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== primitives (conv incl. split-K 128-wide tiles, gram + lsq fp64 MFMA, GEMM epilogues)"
timeout 400 python -m pytest tests/test_gpu_primitives.py -m gpu -q -s -k "conv3x3 or gram_and_lsq or gemm_geglu or gemm_residual" \
  > "$O/${TAG}_primitives.log" 2>&1
echo "rc=$?"; tail -3 "$O/${TAG}_primitives.log"; grep -E "^FAILED|^E  " "$O/${TAG}_primitives.log" | head -10
echo "== lsq VALU vs MFMA (same inputs)"
timeout 120 python - > "$O/${TAG}_lsq_mfma_vs_valu.txt" 2>&1 <<'EOF'
import os, sys, torch
sys.path.insert(0, os.getcwd()); import mos_path
from mixofshow.hip import ops
def timed(fn, it=50):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it
print(f"{'Cout x Cin':16s} {'VALU us':>9s} {'MFMA us':>9s} {'MFMA TFLOP/s':>13s} {'max rel diff grad':>18s} {'rel diff loss':>14s}")
for cout, cin in ((768, 768), (320, 768), (640, 768), (1280, 768), (320, 320), (640, 640), (1280, 1280)):
    g = torch.Generator().manual_seed(1)
    X = torch.randn(4096, cin, generator=g, dtype=torch.float64).cuda()
    W = (torch.randn(cout, cin, generator=g, dtype=torch.float64) * 0.05).cuda()
    G = X.T @ X; P = (torch.randn(cout, cin, generator=g, dtype=torch.float64).cuda() + W) @ G; c = torch.tensor([1e6], dtype=torch.float64).cuda()
    res = {}
    for mode in ('0', '1'):
        os.environ['MOS_LSQ_MFMA'] = mode
        loss, grad = ops.lsq_loss_grad(W, G, P, c, 4096.0 * cout)
        res[mode] = (loss.clone(), grad.clone(), timed(lambda: ops.lsq_loss_grad(W, G, P, c, 4096.0 * cout)))
    dg = ((res['0'][1] - res['1'][1]).abs().max() / res['0'][1].abs().max()).item()
    dl = abs(res['0'][0].item() - res['1'][0].item()) / abs(res['0'][0].item())
    print(f"{f'{cout} x {cin}':16s} {res['0'][2]:9.1f} {res['1'][2]:9.1f} {2.0 * cout * cin * cin / res['1'][2] / 1e6:13.2f} {dg:18.2e} {dl:14.2e}")
EOF
cat "$O/${TAG}_lsq_mfma_vs_valu.txt"
echo "== conv kernel bench (own vs MIOpen)"
timeout 300 python tools/bench_kernels.py --only conv --iters 30 > "$O/${TAG}_kernel_bench_conv.txt" 2>&1
grep -v "^JSON\|^kernel  " "$O/${TAG}_kernel_bench_conv.txt" | head -30
echo "== end to end: fusion on the device (per-concept Gram batches), smoke, training parity"
timeout 500 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --durations=6 \
  -k "graft_smoke or training_steps_match or train_step_through_vae or one_sd15_level0 or fusion_feature_collection or gradient_fusion_end_to_end or quasi_newton" \
  > "$O/${TAG}_e2e.log" 2>&1
echo "rc=$?"; grep -E "^\[parity\]|passed|failed|Error" "$O/${TAG}_e2e.log" | cut -c1-300 | tail -14
echo "== same-box A/B (defaults = conv threshold 0 + 128-wide split-K + batched time projections + FF2 rule)"
timeout 400 python tools/ab_switches.py --half train --kernels conv3x3,gemm_nt,groupnorm_fused,groupnorm_apply \
  "" "MOS_CONV_SPLITK=0" "MOS_FF2_OWN=0" > "$O/${TAG}_ab_train.txt" 2>&1
cut -c1-160 "$O/${TAG}_ab_train.txt"
timeout 500 python tools/ab_switches.py --half regional --timeout 300 --kernels conv3x3,gemm_nt,groupnorm_fused,groupnorm_apply,attn_fwd \
  "" "MOS_CONV_SPLITK=0" "MOS_FF2_OWN=0" > "$O/${TAG}_ab_regional.txt" 2>&1
cut -c1-160 "$O/${TAG}_ab_regional.txt"
echo "== configs[3] fusion (fp64 MFMA closure, per-concept Gram batches)"
python bench.py --mode fusion --concepts 14 --steps 1 --warmup 1 --no-cpu-baseline > "$O/${TAG}_bench_fusion.json" 2> "$O/${TAG}_bench_fusion.err"
echo "rc=$?"; grep "fusion pass" "$O/${TAG}_bench_fusion.err"; cut -c1-200 "$O/${TAG}_bench_fusion.json"
