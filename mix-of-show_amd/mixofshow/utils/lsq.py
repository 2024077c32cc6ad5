"""Layer-wise least squares of gradient fusion on the Gram form (HIP kernels mos_gram_accumulate /
mos_lsq_loss_grad_gram).

Reference: gradient_fusion.py:22-96. `update_quasi_newton(K_target, V_target, W, iters, device)` minimises
mean((K W^T - V)^2) from W with ONE torch.optim.LBFGS.step (lr 1, history 25, strong Wolfe, tolerances 1e-16) and
returns the best-loss iterate; its closure re-uploads every 5000-row chunk of K and V from the host on every
function evaluation. Here the data is reduced ONCE to G = K^T K, P = V^T K, c = sum V^2 (streamed through the MFMA
Gram kernel, fp64 accumulators) and the same L-BFGS runs on the equivalent Gram-form loss
    L(W) = (tr(W G W^T) - 2 tr(W P^T) + c) / (n * Cout),   dL/dW = 2 (W G - P) / (n * Cout)
evaluated in fp64 on the device — same algorithm, same hyper-parameters, same best-iterate rule; the optimiser loop is
mixofshow.utils.lbfgs (torch.optim.LBFGS's iteration without its ~60 device read-backs per iteration).
"""
import torch

from mixofshow.hip import ops


class GramAccumulator:
    """Running Gram statistics of one layer; `add(X, Y)` may be called from forward hooks / feature taps."""

    def __init__(self, cin, cout, device):
        self.G = torch.zeros(cin, cin, dtype=torch.float64, device=device)
        self.P = torch.zeros(cout, cin, dtype=torch.float64, device=device)
        self.c = torch.zeros(1, dtype=torch.float64, device=device)
        self.n = 0
        self.cin, self.cout = cin, cout

    @staticmethod
    def _rows(t, channels):
        if t.dim() == 4:  # conv features (b, C, h, w) -> (b*h*w, C)
            t = t.permute(0, 2, 3, 1)
        return t.reshape(-1, channels)

    def add(self, X, Y, exact_fp32=False):
        """X (.., Cin), Y (.., Cout) on the device. Half inputs go straight to the kernel. fp32 inputs that are not
        half-representable are split x = hi + lo (two half tensors) so that the Gram products keep ~22 bits."""
        X, Y = self._rows(X, self.cin), self._rows(Y, self.cout)
        self.n += X.shape[0]
        if X.dtype in (torch.float16, torch.bfloat16) and Y.dtype == X.dtype:
            ops.gram_accumulate(X.contiguous(), Y.contiguous(), self.G, self.P, self.c)
            return
        X, Y = X.float(), Y.float()
        if not exact_fp32:
            ops.gram_accumulate(X.half().contiguous(), Y.half().contiguous(), self.G, self.P, self.c)
            return
        xh, yh = X.half(), Y.half()
        if bool(((xh.float() == X).all() & (yh.float() == Y).all()).item()):   # recorded from an fp16 pipeline
            ops.gram_accumulate(xh.contiguous(), yh.contiguous(), self.G, self.P, self.c)
            return
        xl, yl = (X - xh.float()).half(), (Y - yh.float()).half()
        # [hi | lo] stacked along channels: one kernel call yields all cross terms
        X2 = torch.cat([xh, xl], 1).contiguous()
        Y2 = torch.cat([yh, yl], 1).contiguous()
        ci, co = self.cin, self.cout
        G2 = torch.zeros(2 * ci, 2 * ci, dtype=torch.float64, device=X.device)
        P2 = torch.zeros(2 * co, 2 * ci, dtype=torch.float64, device=X.device)
        c2 = torch.zeros(1, dtype=torch.float64, device=X.device)
        ops.gram_accumulate(X2, Y2, G2, P2, c2)
        self.G += G2[:ci, :ci] + G2[:ci, ci:] + G2[ci:, :ci] + G2[ci:, ci:]
        self.P += P2[:co, :ci] + P2[:co, ci:] + P2[co:, :ci] + P2[co:, ci:]
        self.c += (Y.double()**2).sum()


def lbfgs_on_gram(W0, acc, iters, solver='lean'):
    """L-BFGS of the reference (gradient_fusion.py:78-85) on the Gram-form loss. W0 (Cout, Cin) any float dtype.
    Returns the best-loss iterate over all function evaluations as fp32 on the CPU (like the reference, :72-74,96).
    solver='lean': mixofshow.utils.lbfgs.minimize (same algorithm, two host read-backs per iteration);
    solver='torch': torch.optim.LBFGS itself (~60 read-backs per iteration) — kept as the yardstick of the tests."""
    dev = acc.G.device
    nm = float(acc.n) * acc.cout
    best = {'loss': float('inf'), 'W': None}
    shape = (acc.cout, acc.cin)

    if solver == 'lean':
        from mixofshow.utils import lbfgs

        def value_and_grad(x):
            loss, grad = ops.lsq_loss_grad(x.view(shape), acc.G, acc.P, acc.c, nm)
            return loss, grad.reshape(-1)

        def on_eval(x, lv):
            if lv < best['loss']:
                best['loss'], best['W'] = lv, x          # evaluation points are never modified afterwards

        x0 = W0.detach().to(dev, torch.float64).reshape(-1).contiguous().clone()
        lbfgs.minimize(value_and_grad, x0, iters, history_size=25, lr=1.0, tolerance_grad=1e-16, tolerance_change=1e-16,
                       on_eval=on_eval)
        return best['W'].view(shape).to(torch.float32).cpu(), best['loss']

    W = W0.detach().to(dev, torch.float64).clone().requires_grad_(True)

    def closure():
        opt.zero_grad()
        loss, grad = ops.lsq_loss_grad(W.detach().contiguous(), acc.G, acc.P, acc.c, nm)
        W.grad = grad
        lv = float(loss)
        if lv < best['loss']:
            best['loss'] = lv
            best['W'] = W.detach().clone()
        return loss

    opt = torch.optim.LBFGS([W], lr=1, max_iter=iters, history_size=25, line_search_fn='strong_wolfe',
                            tolerance_grad=1e-16, tolerance_change=1e-16)
    opt.step(closure)
    return best['W'].to(torch.float32).cpu(), best['loss']


def lbfgs_on_gram_many(W0s, accs, iters):
    """`lbfgs_on_gram(W0, acc, iters)` for several independent layers at once, advanced in lock step by
    mixofshow.utils.lbfgs.minimize_many: ONE host read-back per round for all layers instead of one per layer and request.
    Every layer runs exactly the operations of its own `lbfgs_on_gram` call (bit-identical results); all of them live on the
    device of the statistics and are issued on the current stream. Returns [(W fp32 on the CPU, best loss), ...]."""
    from mixofshow.utils import lbfgs
    bests, gens, shapes = [], [], []
    for W0, acc in zip(W0s, accs):
        dev = acc.G.device
        nm = float(acc.n) * acc.cout
        shape = (acc.cout, acc.cin)
        best = {'loss': float('inf'), 'W': None}

        def value_and_grad(x, acc=acc, shape=shape, nm=nm):
            loss, grad = ops.lsq_loss_grad(x.view(shape), acc.G, acc.P, acc.c, nm)
            return loss, grad.reshape(-1)

        def on_eval(x, lv, best=best):
            if lv < best['loss']:
                best['loss'], best['W'] = lv, x          # evaluation points are never modified afterwards

        x0 = W0.detach().to(dev, torch.float64).reshape(-1).contiguous().clone()
        gens.append(lbfgs.minimize_steps(value_and_grad, x0, iters, history_size=25, lr=1.0, tolerance_grad=1e-16,
                                         tolerance_change=1e-16, on_eval=on_eval))
        bests.append(best)
        shapes.append(shape)
    lbfgs.minimize_many(gens)
    return [(b['W'].view(sh).to(torch.float32).cpu(), b['loss']) for b, sh in zip(bests, shapes)]


def lbfgs_direct_form(K_target, V_target, W, iters, device, chunk=5000):
    """PARITY / DIAGNOSTIC mode (not the fusion path): the same optimiser loop (mixofshow.utils.lbfgs) on the reference's
    own closure arithmetic -- fp32, loss = mean((K W^T - V)^2) summed over 5000-row chunks as chunk_compute_mse does
    (gradient_fusion.py:22-35), gradient by autograd like the reference's loss.backward() -- instead of the fp64 Gram form. Two truncated
    L-BFGS runs on an ill-conditioned layer only agree while their closures round alike; this mode separates "Gram
    form vs direct form" from "different features" when fused weights are compared with the reference's
    (tests/test_fusion_cpu.py). Linear layers only (2-D W)."""
    from mixofshow.utils import lbfgs
    assert W.dim() == 2
    K = K_target.detach().to(device, torch.float32)
    V = V_target.detach().to(device, torch.float32)
    n, cout = K.shape[0], W.shape[0]
    best = {'loss': float('inf'), 'W': None}

    def value_and_grad(x):
        Wm = x.detach().view(W.shape).clone().requires_grad_(True)
        with torch.enable_grad():
            loss = 0
            for s0 in range(0, n, chunk):      # chunk_compute_mse: F.mse_loss(F.linear(K, W), V) * rows, summed, / n
                k, v = K[s0:s0 + chunk], V[s0:s0 + chunk]
                loss = loss + torch.nn.functional.mse_loss(torch.nn.functional.linear(k, Wm), v) * k.shape[0]
            loss = loss / n
            (grad, ) = torch.autograd.grad(loss, Wm)
        return loss.detach(), grad.reshape(-1)

    def on_eval(x, lv):
        if lv < best['loss']:
            best['loss'], best['W'] = lv, x

    x0 = W.detach().to(K.device, torch.float32).reshape(-1).contiguous().clone()
    lbfgs.minimize(value_and_grad, x0, iters, history_size=25, lr=1.0, tolerance_grad=1e-16, tolerance_change=1e-16,
                   on_eval=on_eval)
    return best['W'].view(W.shape).to(torch.float32).cpu(), best['loss']


def update_quasi_newton(K_target, V_target, W, iters, device, form='gram'):
    """Drop-in for the reference's update_quasi_newton (gradient_fusion.py:38-96); K/V may live on the CPU,
    W is 2-D (Linear) or 4-D (1x1 conv, features (n, C, h, w)). form='direct': the parity mode above."""
    if form == 'direct':
        return lbfgs_direct_form(K_target, V_target, W, iters, device)[0]
    assert form == 'gram', form
    conv = W.dim() == 4
    cout, cin = W.shape[0], W.shape[1]
    acc = GramAccumulator(cin, cout, device)
    n = K_target.shape[0]
    chunk = 65536 if not conv else 64
    for s in range(0, n, chunk):
        acc.add(K_target[s:s + chunk].to(device), V_target[s:s + chunk].to(device), exact_fp32=True)
    Wn, loss = lbfgs_on_gram(W.reshape(cout, cin), acc, iters)
    return Wn.reshape(W.shape)
