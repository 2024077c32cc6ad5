"""Gradient-fusion solver on one layer: mixofshow.utils.lbfgs (lean) vs torch.optim.LBFGS on the same Gram problem.
  python tools/bench_lbfgs.py            # GPU box; prints wall seconds, evaluations, best loss per solver"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mos_path  # noqa: E402,F401
import torch  # noqa: E402

from mixofshow.hip import ops  # noqa: E402
from mixofshow.utils import lsq  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    cases = (('warm-up', 64, 64, 32, 20), ('cross-attn to_k, n = 84 rows (rank-deficient: runs all 500 iterations)', 768, 320, 84, 500),
             ('CLIP q_proj, n = 462 rows', 768, 768, 462, 500), ('spatial to_q L0, n = 81920', 320, 320, 81920, 50))
    if len(sys.argv) > 1:
        cases = cases[:int(sys.argv[1])]
    for name, cin, cout, n, iters in cases:
        X = torch.randn(n, cin, generator=g)
        Y = X @ (torch.randn(cout, cin, generator=g) * 0.05).T + 0.01 * torch.randn(n, cout, generator=g)
        acc = lsq.GramAccumulator(cin, cout, dev)
        for s in range(0, n, 16384):
            acc.add(X[s:s + 16384].to(dev).half(), Y[s:s + 16384].to(dev).half())
        W0 = torch.zeros(cout, cin)
        for solver in ('lean', 'torch'):
            calls = [0]
            real = ops.lsq_loss_grad

            def counted(*a):
                calls[0] += 1
                return real(*a)

            ops.lsq_loss_grad = counted
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, loss = lsq.lbfgs_on_gram(W0, acc, iters, solver=solver)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ops.lsq_loss_grad = real
            print(f'{name:72s} {cout}x{cin} iters {iters:3d}  {solver:5s}: {dt:7.3f} s, {calls[0]:4d} evaluations, '
                  f'{dt / max(1, calls[0]) * 1e3:6.3f} ms/eval, best loss {loss:.6e}', flush=True)


if __name__ == '__main__':
    main()
