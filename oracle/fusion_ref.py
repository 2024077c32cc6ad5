"""ORACLE — test infrastructure. CPU restatement of the gradient-fusion least squares of the reference
(gradient_fusion.py:22-96): loss = mean((X W^T - Y)^2) in 5000-row chunks, minimised from the
pretrained W by ONE torch.optim.LBFGS.step (lr 1, history 25, strong Wolfe, tolerances 1e-16),
returning the best-loss iterate seen by any closure evaluation."""
import torch
import torch.nn.functional as F


def chunk_compute_mse_ref(K_target, V_target, W, chunk_size=5000):
    n = K_target.size(0)
    loss = 0
    for s in range(0, n, chunk_size):
        e = min(s + chunk_size, n)
        loss = loss + F.mse_loss(F.linear(K_target[s:e], W), V_target[s:e]) * (e - s)
    return loss / n


def update_quasi_newton_ref(K_target, V_target, W, iters):
    W = W.detach().clone().requires_grad_(True)
    K_target, V_target = K_target.detach(), V_target.detach()
    best = {'loss': float('inf'), 'W': None}

    def closure():
        opt.zero_grad()
        if W.dim() == 4:
            loss = F.mse_loss(F.conv2d(K_target, W), V_target)
        else:
            loss = chunk_compute_mse_ref(K_target, V_target, W)
        if loss < best['loss']:
            best['loss'] = loss.detach().clone()
            best['W'] = W.detach().clone()
        loss.backward()
        return loss

    opt = torch.optim.LBFGS([W], lr=1, max_iter=iters, history_size=25, line_search_fn='strong_wolfe',
                            tolerance_grad=1e-16, tolerance_change=1e-16)
    opt.step(closure)
    return best['W']


def lsq_loss_ref(K_target, V_target, W):
    if W.dim() == 4:
        return F.mse_loss(F.conv2d(K_target, W), V_target)
    return F.mse_loss(F.linear(K_target, W), V_target)
