#!/usr/bin/env bash
# Round-4, fourth GPU call: the software-pipelined d = 40 attention forward, the restored 64 x 64 split-K plan, the prefetching
# fp64-MFMA closure.                                            bash tools/r04_validate_attention.sh [tag]
set -u
TAG="${1:-r04d}"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
O="$ROOT/gpurun_out"; mkdir -p "$O"
echo "== primitives (attention incl. the pipelined forward, conv, gram + lsq)"
timeout 500 python -m pytest tests/test_gpu_primitives.py -m gpu -q -s -k "attention or conv3x3 or gram_and_lsq" \
  > "$O/${TAG}_primitives.log" 2>&1
echo "rc=$?"; tail -3 "$O/${TAG}_primitives.log"; grep -E "^FAILED|^E  " "$O/${TAG}_primitives.log" | head -10
grep "pipelined" "$O/${TAG}_primitives.log" | head -12
echo "== attention kernel bench, pipelined forward on / off"
for p in 1 0; do
  MOS_ATTN_PIPE=$p timeout 200 python tools/bench_kernels.py --only attn --iters 20 > "$O/${TAG}_kernel_bench_attn_pipe${p}.txt" 2>&1
  echo "-- MOS_ATTN_PIPE=$p"; grep -E "^attn_fwd f16 d40" "$O/${TAG}_kernel_bench_attn_pipe${p}.txt"
done
echo "== lsq VALU vs MFMA (prefetching)"
timeout 120 python - > "$O/${TAG}_lsq_mfma_vs_valu.txt" 2>&1 <<'PY'
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
PY
cat "$O/${TAG}_lsq_mfma_vs_valu.txt"
echo "== end to end: sampling parity with the pipelined forward (teacher-forced EDLoRA sd15), training step vs twin"
timeout 420 python -m pytest tests/test_gpu_end_to_end.py -m gpu -q -s --durations=5 \
  -k "edlora_sd15_hot_path or train_step_through_vae or graft_smoke" > "$O/${TAG}_e2e.log" 2>&1
echo "rc=$?"; grep -E "^\[parity\]|passed|failed|Error" "$O/${TAG}_e2e.log" | cut -c1-330 | tail -8
echo "== same-box A/B"
timeout 400 python tools/ab_switches.py --half train --kernels conv3x3,attn_fwd,attn_bwd_dkdv,attn_bwd_dq \
  "" "MOS_ATTN_PIPE=0" "MOS_CONV_SPLIT_TILE=128" > "$O/${TAG}_ab_train.txt" 2>&1
cut -c1-170 "$O/${TAG}_ab_train.txt"
timeout 500 python tools/ab_switches.py --half regional --timeout 300 --kernels conv3x3,attn_fwd,gemm_nt,groupnorm_fused \
  "" "MOS_ATTN_PIPE=0" "MOS_CONV_SPLIT_TILE=128" > "$O/${TAG}_ab_regional.txt" 2>&1
cut -c1-170 "$O/${TAG}_ab_regional.txt"
