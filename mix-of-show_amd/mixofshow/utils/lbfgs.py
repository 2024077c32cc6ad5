"""L-BFGS with strong-Wolfe line search for a device-resident objective, without the host round trips of
`torch.optim.LBFGS`.

The reference's gradient fusion (gradient_fusion.py:78-85) runs ONE `torch.optim.LBFGS(lr=1, history_size=25,
line_search_fn='strong_wolfe', tolerance_* = 1e-16).step(closure)` per layer. That optimiser is the algorithm to
reproduce; its implementation, though, synchronises with the device ~2 x history + 6 times per iteration (every
`q.add_(old_dirs[i], alpha=-al[i])` of the two-loop recursion reads a 0-dim tensor back as a Python scalar, and so does
every comparison in the line search). With the Gram-form closure a function evaluation is one ~0.2 ms kernel, so
fourteen-concept fusion spent ~100 of its 132 s in those round trips.

Here the same iteration is organised so that the host reads back ONE small vector per function evaluation
([loss, g.d, max|g|]) and one per search direction ([g.d, max|d|], plus sum|g| for the first):
  * the two-loop recursion is replaced by its closed form, the compact representation of the L-BFGS inverse Hessian
    (Byrd, Nocedal, Schnabel 1994, eq. 3.1):  H = gamma I + [S  gamma Y] M [S^T ; gamma Y^T],
    M = [[R^-T (D + gamma Y^T Y) R^-1, -R^-T], [-R^-1, 0]],  R = triu(S^T Y),  D = diag(S^T Y) —
    two (k x n) mat-vecs, two k x k triangular solves and one (n x k) combination, all on the device, identical to the
    recursion in exact arithmetic (same pairs, same gamma = s.y / y.y of the newest accepted pair);
  * curvature s.y of a new pair needs no reduction: y.s = t (g_new.d - g_old.d), both already on the host;
  * the line search (`_strong_wolfe` below) is the optimiser's bracketing / zoom procedure on Python floats.
Update rules, constants (c1 = 1e-4, c2 = 0.9, curvature threshold 1e-10, first step min(1, 1/|g|_1), max_eval =
5/4 max_iter, max_ls = max_eval - evals so far), termination tests and their order follow torch.optim.LBFGS.step, so the
iterates agree with it to rounding (tests/test_fusion_cpu.py compares them on random and on the reference's golden
problems).

The iteration is written as a GENERATOR (`minimize_steps`): wherever the host needs device values it yields the small device
vector and is sent back its values as a Python list. `minimize` drives one problem (read-back = `.tolist()`); `minimize_many`
drives any number of independent problems in lock step on one stream and answers ALL their pending requests with ONE
concatenation and ONE read-back per round -- the layers of a fusion are independent problems, and the host round trip, not
the arithmetic, is what an iteration costs. Each problem executes exactly the operations of its own
sequential run, in the same order: the iterates are bit-identical.
"""
import math

import torch

def _cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    """Minimiser of the cubic through (x1, f1, g1), (x2, f2, g2), clipped to `bounds` (default: the interval)."""
    if bounds is not None:
        lo, hi = bounds
    else:
        lo, hi = (x1, x2) if x1 <= x2 else (x2, x1)
    d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2)
    d2_square = d1 * d1 - g1 * g2
    if d2_square >= 0:
        d2 = math.sqrt(d2_square)
        if x1 <= x2:
            pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2))
        else:
            pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2))
        return min(max(pos, lo), hi)
    return (lo + hi) / 2.0


def _strong_wolfe(evaluate, t, d_norm, f, g, gtd, c1=1e-4, c2=0.9, tolerance_change=1e-9, max_ls=25):
    """Bracketing + zoom line search for the strong Wolfe conditions (a generator, see the module docstring: use with
    `yield from`). `evaluate(t)` is a generator returning (f(t), g(t), g(t).d) with f and g.d Python floats and g a device
    tensor that is not modified afterwards. Returns (f, g, t, evaluations, g.d)."""
    f_new, g_new, gtd_new = yield from evaluate(t)
    evals = 1
    t_prev, f_prev, g_prev, gtd_prev = 0.0, f, g, gtd
    done = False
    ls_iter = 0
    bracket = bracket_f = bracket_g = bracket_gtd = None
    while ls_iter < max_ls:
        if f_new > (f + c1 * t * gtd) or (ls_iter > 1 and f_new >= f_prev):
            bracket, bracket_f, bracket_g, bracket_gtd = [t_prev, t], [f_prev, f_new], [g_prev, g_new], [gtd_prev, gtd_new]
            break
        if abs(gtd_new) <= -c2 * gtd:
            bracket, bracket_f, bracket_g, bracket_gtd = [t], [f_new], [g_new], [gtd_new]
            done = True
            break
        if gtd_new >= 0:
            bracket, bracket_f, bracket_g, bracket_gtd = [t_prev, t], [f_prev, f_new], [g_prev, g_new], [gtd_prev, gtd_new]
            break
        min_step = t + 0.01 * (t - t_prev)
        max_step = t * 10
        tmp = t
        t = _cubic_interpolate(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, bounds=(min_step, max_step))
        t_prev, f_prev, g_prev, gtd_prev = tmp, f_new, g_new, gtd_new
        f_new, g_new, gtd_new = yield from evaluate(t)
        evals += 1
        ls_iter += 1
    if ls_iter == max_ls:
        bracket, bracket_f, bracket_g, bracket_gtd = [0.0, t], [f, f_new], [g, g_new], [gtd, gtd_new]

    insuf_progress = False
    low_pos, high_pos = (0, 1) if bracket_f[0] <= bracket_f[-1] else (1, 0)
    while not done and ls_iter < max_ls:
        if abs(bracket[1] - bracket[0]) * d_norm < tolerance_change:
            break
        t = _cubic_interpolate(bracket[0], bracket_f[0], bracket_gtd[0], bracket[1], bracket_f[1], bracket_gtd[1])
        # too close to an end of the bracket twice in a row, or on it: step 10 % of the bracket inside instead
        eps = 0.1 * (max(bracket) - min(bracket))
        if min(max(bracket) - t, t - min(bracket)) < eps:
            if insuf_progress or t >= max(bracket) or t <= min(bracket):
                t = max(bracket) - eps if abs(t - max(bracket)) < abs(t - min(bracket)) else min(bracket) + eps
                insuf_progress = False
            else:
                insuf_progress = True
        else:
            insuf_progress = False
        f_new, g_new, gtd_new = yield from evaluate(t)
        evals += 1
        ls_iter += 1
        if f_new > (f + c1 * t * gtd) or f_new >= bracket_f[low_pos]:
            bracket[high_pos], bracket_f[high_pos], bracket_g[high_pos], bracket_gtd[high_pos] = t, f_new, g_new, gtd_new
            low_pos, high_pos = (0, 1) if bracket_f[0] <= bracket_f[1] else (1, 0)
        else:
            if abs(gtd_new) <= -c2 * gtd:
                done = True
            elif gtd_new * (bracket[high_pos] - bracket[low_pos]) >= 0:
                bracket[high_pos], bracket_f[high_pos] = bracket[low_pos], bracket_f[low_pos]
                bracket_g[high_pos], bracket_gtd[high_pos] = bracket_g[low_pos], bracket_gtd[low_pos]
            bracket[low_pos], bracket_f[low_pos], bracket_g[low_pos], bracket_gtd[low_pos] = t, f_new, g_new, gtd_new
    return bracket_f[low_pos], bracket_g[low_pos], bracket[low_pos], evals, bracket_gtd[low_pos]


class _History:
    """The last `size` accepted (s, y) pairs as rows of S, Y in a ring (the oldest row is overwritten in place: no
    shifting of the (size x n) buffers), with S^T Y and Y^T Y kept up to date, and the next search direction.

    What an iteration costs on the device is the traffic over S and Y (25 rows of Cout*Cin fp64 each: 2 x 118 MB for a
    768 x 768 layer -- the closure itself is an 84 us kernel), so `step` touches them as little as the algorithm allows:
      * the direction needs S g, Y g and one combination each of the rows of S and of Y: FOUR passes (the two-loop
        recursion needs the same four);
      * the inner products of a NEW pair with the stored ones need no pass at all: y = g - g_prev, so
        s_i.y = (S g)_i - (S g_prev)_i and y_i.y = (Y g)_i - (Y g_prev)_i with both vectors already computed for the
        directions of this and of the previous iteration (kept in `Sg`, `Yg`); s.y of the new pair is known on the host
        (t (g.d - g_prev.d)), y.y is one dot product. Only the upper triangle of S^T Y (pairs in age order) enters the
        compact representation, so the products s_new.y_i are not needed. (The earlier form spent three more passes on
        S y, Y s, Y y.)
    Every buffer is kept TWICE, back to back (rows r and r + size hold the same pair; the small matrices repeat in all four
    quadrants), so that the pairs in AGE order are always one contiguous window [start, start + k) of the doubled buffer:
    a view, no gathers and no permutation scatters."""

    def __init__(self, size, n, like):
        self.size, self.n = size, n
        kw = dict(dtype=like.dtype, device=like.device)
        self.S = torch.empty(2 * size, n, **kw)
        self.Y = torch.empty(2 * size, n, **kw)
        self.SY = torch.zeros(2 * size, 2 * size, **kw)      # [i, j] = s_i . y_j, valid where pair i is not younger than pair j
        self.YY = torch.zeros(2 * size, 2 * size, **kw)
        self.Sg = torch.zeros(size, **kw)                    # S g, Y g of the previous `step` (physical row order)
        self.Yg = torch.zeros(size, **kw)
        self.k = 0              # pairs stored
        self.start = 0          # physical row of the oldest pair once the ring is full

    def step(self, pair, g, gamma):
        """Optionally store `pair` = (s, y = g - g_prev, s.y as a Python float), then return (-H g, gamma) with H the L-BFGS
        inverse Hessian of the stored pairs and H0 = gamma I, gamma = s.y / y.y of the newest pair (a device scalar)."""
        size = self.size
        slot = None
        if pair is not None:
            s, y, ys = pair
            if self.k == size:
                slot = self.start
                self.start = (self.start + 1) % size
            else:
                slot = self.k
                self.k += 1
            self.S.view(2, size, self.n)[:, slot] = s
            self.Y.view(2, size, self.n)[:, slot] = y
            yy_new = y.dot(y)
            gamma = ys / yy_new
        k = self.k
        if k == 0:
            return g.neg() * gamma, gamma
        # (two streaming HIP kernels for these passes were measured in round 5 and LOST to the four skinny rocBLAS dgemv calls:
        #  one 14-concept fusion 46.0 s against 36.4 s, profiles/r05c1_fusion_ab.txt -- removed)
        Sg, Yg = self.S[:k] @ g, self.Y[:k] @ g          # physical row order, the new pair included
        if slot is not None:
            sy_col, yy_col = Sg - self.Sg[:k], Yg - self.Yg[:k]           # (entry `slot` is meaningless: set below)
            sy_col[slot] = ys
            yy_col[slot] = yy_new
            SY4, YY4 = self.SY.view(2, size, 2, size), self.YY.view(2, size, 2, size)
            SY4[:, :k, :, slot] = sy_col.view(1, k, 1)
            YY4[:, :k, :, slot] = yy_col.view(1, k, 1)
            YY4[:, slot, :, :k] = yy_col.view(1, 1, k)
        self.Sg[:k] = Sg
        self.Yg[:k] = Yg
        lo = self.start if k == size else 0                  # the age-ordered window of the doubled buffers
        if lo:
            Sg, Yg = Sg.repeat(2)[lo:lo + k], Yg.repeat(2)[lo:lo + k]
        S, Y = self.S[lo:lo + k], self.Y[lo:lo + k]
        SY, YY = self.SY[lo:lo + k, lo:lo + k], self.YY[lo:lo + k, lo:lo + k]
        # compact representation (module docstring) with q = -g:  -H g = gamma (Y^T u - g) + S^T v,
        #   u = R^-1 S g,   v = R^-T (gamma Y g - (D + gamma Y^T Y) u),   R = triu(S^T Y), D = diag(S^T Y)
        R = torch.triu(SY)
        u = torch.linalg.solve_triangular(R, Sg.unsqueeze(1), upper=True).squeeze(1)
        M = YY * gamma
        M.diagonal().add_(SY.diagonal())
        mid = torch.addmv(Yg * gamma, M, u, alpha=-1)
        v = torch.linalg.solve_triangular(R.t(), mid.unsqueeze(1), upper=False).squeeze(1)
        out = torch.addmv(g, Y.t(), u, beta=-1)
        out *= gamma
        return torch.addmv(out, S.t(), v), gamma


def _absmax(v):
    return torch.linalg.vector_norm(v, float('inf'))


def minimize_steps(value_and_grad, x0, max_iter, history_size=25, lr=1.0, tolerance_grad=1e-16, tolerance_change=1e-16,
                   max_eval=None, on_eval=None):
    """The generator behind `minimize` / `minimize_many`: yields a small 1-D device tensor whenever the host needs its
    values and expects them back (`send`) as a list of Python floats; returns (x, f(x), evaluations)."""
    max_eval = max_iter * 5 // 4 if max_eval is None else max_eval
    x = x0

    def evaluate_at(xt, d=None):
        f_t, g_t = value_and_grad(xt)
        parts = [f_t.reshape(()), _absmax(g_t)]
        if d is not None:
            parts.append(g_t.dot(d))
        vals = yield torch.stack(parts)              # the one host read-back of this evaluation
        if on_eval is not None:
            on_eval(xt, vals[0])
        return vals, g_t

    (loss, gmax), g = yield from evaluate_at(x)
    evals = 1
    if gmax <= tolerance_grad:
        return x, loss, evals
    hist = _History(history_size, x.numel(), x)
    gamma = 1.0
    d = t = None
    gtd_new = None
    n_iter = 0
    while n_iter < max_iter:
        n_iter += 1
        if n_iter == 1:
            d = g.neg()
        else:
            ys = t * (gtd_new - gtd)                 # y.s = (g_new - g_old).(t d)
            pair = (d if t == 1.0 else d * t, g - prev_g, ys) if ys > 1e-10 else None
            d, gamma = hist.step(pair, g, gamma)     # (gamma stays on the device)
        prev_g, prev_loss = g, loss
        if n_iter == 1:                              # (|g|_1 only scales the very first step)
            gtd, d_norm, g_l1 = yield torch.stack([g.dot(d), _absmax(d), torch.linalg.vector_norm(g, 1)])
            t = min(1.0, 1.0 / g_l1) * lr
        else:
            gtd, d_norm = yield torch.stack([g.dot(d), _absmax(d)])
            t = lr
        if gtd > -tolerance_change:
            break
        x_init = x

        def evaluate(step):
            xt = torch.add(x_init, d, alpha=step)
            (f_t, gmax_t, gtd_t), g_t = yield from evaluate_at(xt, d)
            trial[step] = (xt, gmax_t)
            return f_t, g_t, gtd_t

        trial = {}
        loss, g, t, ls_evals, gtd_new = yield from _strong_wolfe(evaluate, t, d_norm, loss, g, gtd, tolerance_change=1e-9,
                                                                 max_ls=max_eval - evals)
        x, gmax = trial[t] if t in trial else (x_init, gmax)       # t == 0: the search returned the starting point
        evals += ls_evals
        if n_iter == max_iter or evals >= max_eval:
            break
        if gmax <= tolerance_grad:
            break
        if d_norm * abs(t) <= tolerance_change:
            break
        if abs(loss - prev_loss) < tolerance_change:
            break
    return x, loss, evals


def minimize(value_and_grad, x0, max_iter, history_size=25, lr=1.0, tolerance_grad=1e-16, tolerance_change=1e-16,
             max_eval=None, on_eval=None):
    """Minimise f from x0 (1-D device tensor). `value_and_grad(x)` -> (f(x) as a 0-dim device tensor, grad f(x) as a
    1-D tensor that the caller does not modify afterwards). `on_eval(x, f_float)` is called after every evaluation
    (the reference keeps the best-loss iterate over ALL evaluations, line-search trials included).
    Returns (x, f(x), evaluations)."""
    gen = minimize_steps(value_and_grad, x0, max_iter, history_size, lr, tolerance_grad, tolerance_change, max_eval, on_eval)
    try:
        req = next(gen)
        while True:
            req = gen.send(req.tolist())
    except StopIteration as stop:
        return stop.value


def minimize_many(problems):
    """`problems`: generators made by `minimize_steps` (independent problems on ONE device / stream). Advances all of them in
    lock step; per round the pending requests of all unfinished problems are concatenated and read back together (one
    synchronisation per round instead of one per problem and request). Returns their results in order."""
    gens = list(problems)
    results = [None] * len(gens)
    pending = {}
    for i, gen in enumerate(gens):
        try:
            pending[i] = next(gen)
        except StopIteration as stop:
            results[i] = stop.value
    while pending:
        keys = list(pending)
        reqs = [pending[k].reshape(-1) for k in keys]
        host = (torch.cat(reqs) if len(reqs) > 1 else reqs[0]).tolist()
        off = 0
        for k, r in zip(keys, reqs):
            vals = host[off:off + r.numel()]
            off += r.numel()
            try:
                pending[k] = gens[k].send(vals)
            except StopIteration as stop:
                results[k] = stop.value
                del pending[k]
    return results
