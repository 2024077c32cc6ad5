"""ORACLE — test infrastructure. Torch (CPU or GPU) emulation of the kernel-backed primitives of
mixofshow.hip.ops with the SAME signatures. Two uses, both in tests only:
  * `emulated_hip` fixture: run the product's host orchestration (autograd Functions, processors, trainer)
    on CPU and compare it with the oracle restatement of the reference;
  * GPU parity tests: each HIP primitive against its emulation on the same inputs.
Emulations compute in fp32 from the (possibly half) inputs and round the result to the output dtype once.
"""
import torch

EMULATED = ('lora_pack', 'lora_down', 'linear_fwd', 'linear_fwd_ex', 'linear_bwd', 'linear_fused_fwd', 'linear_fused_bwd', 'attn_fwd', 'attn_bwd', 'attn_probs', 'attn_pv', 'attn_probs_bwd', 'attn_pv_bwd', 'region_attn_fwd',
            'gram_accumulate', 'lsq_loss_grad', 'groupnorm_silu_fwd', 'groupnorm_silu_bwd', 'layernorm_fwd', 'layernorm_bwd', 'add_layernorm_fwd', 'add_layernorm_bwd',
            'geglu_fwd', 'geglu_bwd', 'quick_gelu_fwd', 'quick_gelu_bwd', 'softmax_rows', 'single_head_attention_nograd', 'conv3x3_nhwc', 'conv3x3_s2_nhwc', 'groupnorm_reads_twice')
PAD = 16


def lora_pack(downs, ups, alphas, K, dtype, device):
    r = downs[0].shape[0]
    N = sum(u.reshape(-1, r).shape[0] for u in ups)
    A16 = torch.zeros(PAD, K, dtype=torch.float32, device=device)
    Bp16 = torch.zeros(N, PAD, dtype=torch.float32, device=device)
    n0 = 0
    for g, (d, u, a) in enumerate(zip(downs, ups, alphas)):
        d = d.detach().reshape(r, K).float()
        u = u.detach().reshape(-1, r).float()
        A16[g * r:(g + 1) * r] = d
        Bp16[n0:n0 + u.shape[0], g * r:(g + 1) * r] = a * u
        n0 += u.shape[0]
    A16, Bp16 = A16.to(dtype), Bp16.to(dtype)
    return A16, A16.t().contiguous(), Bp16, Bp16.t().contiguous()


def lora_down(x, A16):
    return (x.float() @ A16.float().t()).to(x.dtype)


def linear_fwd(x, W, t=None, Bp16=None, bias=None, out=None):
    y = x.float() @ W.float().t()
    if t is not None:
        y = y + t.float() @ Bp16.float().t()
    if bias is not None:
        y = y + bias
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def linear_fused_fwd(x, W, A16, Bp16, bias=None, need_t=True):
    """mos_lora_linear_fused_fwd: t = x A16^T accumulated in fp32, ROUNDED to the half type, then y as linear_fwd."""
    t = (x.float() @ A16.float().t()).to(x.dtype)
    return linear_fwd(x, W, t, Bp16, bias), (t if need_t else None)


def linear_fwd_ex(x, W, A16=None, Bp16=None, bias=None, residual=None, need_t=False):
    """mos_lora_linear_fwd_ex: the GEMM rounded to the half type, then the residual added to the rounded values, rounded again."""
    t = None
    if A16 is not None:
        t = (x.float() @ A16.float().t()).to(x.dtype)
    y = linear_fwd(x, W, t, Bp16, bias)
    if residual is not None:
        y = (y.float() + residual.float()).to(x.dtype)
    return y, (t if need_t else None)


def linear_fused_bwd(dy, x, Wt, t, A16T, BpT, grad_targets, rank, need_dx=True):
    """mos_lora_linear_fused_bwd: dt = dy BpT^T (rounded), dx = dy Wt^T + dt A16T^T, factor gradients written /
    accumulated straight into the per-site fp32 targets (alpha folded into the up gradient)."""
    dyf = dy.float()
    dt = (dyf @ BpT.float().t()).to(dy.dtype)
    dx = None
    if need_dx:
        dx = (dyf @ Wt.float().t() + dt.float() @ A16T.float().t()).to(dy.dtype)
    if grad_targets:
        dA = dt.float().t() @ x.float()          # (16, K)
        dB = t.float().t() @ dyf                 # (16, N)
        n0 = 0
        for g, (dg, ug, alpha, n_rows, acc_d, acc_u) in enumerate(grad_targets):
            if dg is not None:
                v = dA[g * rank:(g + 1) * rank].reshape(dg.shape)
                dg.copy_(dg + v if acc_d else v)
            if ug is not None:
                v = (alpha * dB[g * rank:(g + 1) * rank, n0:n0 + n_rows].t()).reshape(ug.shape)
                ug.copy_(ug + v if acc_u else v)
            n0 += n_rows
    return dx


def linear_bwd(dy, x, Wt, t, A16T, BpT, need_dx=True, need_lora=True, lora_cols=16):
    dyf = dy.float()
    lora = BpT is not None
    dt = (dyf @ BpT.float().t()).to(dy.dtype) if lora else None
    dx = None
    if need_dx:
        dx = dyf @ Wt.float().t()
        if lora:
            dx = dx + dt.float() @ A16T.float().t()
        dx = dx.to(dy.dtype)
    dA16 = dBpT = None
    if lora and need_lora:
        dA16 = dt.float().t() @ x.float()
        dBpT = t.float().t() @ dyf
    return dx, dA16, dBpT


def _heads(t, H):
    B, N, C = t.shape
    return t.float().reshape(B, N, H, C // H).permute(0, 2, 1, 3)


def _causal_mask(s):
    n, m = s.shape[-2], s.shape[-1]
    return torch.ones(n, m, dtype=torch.bool, device=s.device).triu(1)        # key index > query index


def attn_fwd(q, k, v, heads, scale, tok_idx=None, need_lse=True, causal=False):
    B, Nq, C = q.shape
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(_causal_mask(s), float('-inf'))
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - lse[..., None])
    o = (p @ vh).permute(0, 2, 1, 3).reshape(B, Nq, C).to(q.dtype)
    pcols = None
    if tok_idx is not None:
        idx = tok_idx.long()[:, None, None, :].expand(B, heads, Nq, tok_idx.shape[1])
        pcols = torch.gather(p, 3, idx).contiguous()
    return o, (lse.contiguous() if need_lse else None), pcols


def attn_bwd(q, k, v, o, lse, dO, heads, scale, dq, dk, dv, tok_idx=None, pcols=None, dpcols=None, causal=False):
    B, Nq, C = q.shape
    qh, kh, vh, doh = _heads(q, heads), _heads(k, heads), _heads(v, heads), _heads(dO, heads)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(_causal_mask(s), float('-inf'))
    p = torch.exp(s - lse[..., None])
    dvh = p.transpose(-1, -2) @ doh
    dp = doh @ vh.transpose(-1, -2)
    if dpcols is not None:
        idx = tok_idx.long()[:, None, None, :].expand(B, heads, Nq, tok_idx.shape[1])
        dp = dp.scatter_add(3, idx, dpcols.float())
    ds = p * (dp - (p * dp).sum(-1, keepdim=True))
    dqh = (ds @ kh) * scale
    dkh = (ds.transpose(-1, -2) @ qh) * scale

    def back(t):
        return t.permute(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], -1)

    dq.copy_(back(dqh).to(dq.dtype)); dk.copy_(back(dkh).to(dk.dtype)); dv.copy_(back(dvh).to(dv.dtype))
    return dq, dk, dv


def attn_probs(q, k, heads, scale):
    qh, kh = _heads(q, heads), _heads(k, heads)
    p = torch.softmax((qh @ kh.transpose(-1, -2)) * scale, dim=-1)
    return p.reshape(q.shape[0] * heads, q.shape[1], k.shape[1]).to(q.dtype)


def attn_pv(probs, v, heads):
    B, Nkv, C = v.shape
    o = probs.float().reshape(B, heads, -1, Nkv) @ _heads(v, heads)
    return o.permute(0, 2, 1, 3).reshape(B, -1, C).to(v.dtype)


def attn_pv_bwd(probs, v, dO, heads):
    """dP' = dO V^T (dense, layer dtype), dV = P'^T dO: backward of torch.bmm(attention_probs, value) (edlora.py:83)."""
    B, Nkv, C = v.shape
    Pf = probs.float().reshape(B, heads, -1, Nkv)
    dOh, vh = _heads(dO, heads), _heads(v, heads)
    dP = (dOh @ vh.transpose(-1, -2)).reshape(B * heads, -1, Nkv).to(probs.dtype)
    dv = (Pf.transpose(-1, -2) @ dOh).permute(0, 2, 1, 3).reshape(B, Nkv, C).to(v.dtype)
    return dP, dv


def attn_probs_bwd(q, k, probs, dprobs, heads, scale):
    """softmax Jacobian + the two score GEMMs: backward of get_attention_scores (edlora.py:81)."""
    B, Nq, C = q.shape
    Nkv = k.shape[1]
    P, dP = probs.float().reshape(B, heads, Nq, Nkv), dprobs.float().reshape(B, heads, Nq, Nkv)
    dS = P * (dP - (P * dP).sum(-1, keepdim=True)) * scale
    dq = (dS @ _heads(k, heads)).permute(0, 2, 1, 3).reshape(B, Nq, C).to(q.dtype)
    dk = (dS.transpose(-1, -2) @ _heads(q, heads)).permute(0, 2, 1, 3).reshape(B, Nkv, C).to(k.dtype)
    return dq, dk


def region_attn_fwd(q, k_src, v_src, heads, scale, boxes, feat_h, feat_w):
    B, Nq, C = q.shape
    count = torch.zeros(feat_h, feat_w, device=q.device)
    for h0, w0, h1, w1 in boxes:
        count[h0:h1, w0:w1] += 1
    count = count.reshape(-1)
    out = torch.zeros(B, Nq, C, dtype=torch.float32, device=q.device)
    base, _, _ = attn_fwd(q.float(), k_src[0].float(), v_src[0].float(), heads, scale, need_lse=False)
    out[:, count == 0] = base[:, count == 0]
    for r, (h0, w0, h1, w1) in enumerate(boxes):
        inbox = torch.zeros(feat_h, feat_w, dtype=torch.bool, device=q.device)
        inbox[h0:h1, w0:w1] = True
        inbox = inbox.reshape(-1)
        if inbox.any():
            o_r, _, _ = attn_fwd(q.float(), k_src[r + 1].float(), v_src[r + 1].float(), heads, scale, need_lse=False)
            out[:, inbox] += o_r[:, inbox] / count[inbox][None, :, None]
    return out.to(q.dtype)


def gram_accumulate(X, Y, G, P, c):
    Xd, Yd = X.double(), Y.double()
    G += Xd.t() @ Xd
    P += Yd.t() @ Xd
    c += (Yd * Yd).sum()


def lsq_loss_grad(W, G, P, c, n_times_cout):
    R = W @ G - P
    loss = ((R * W).sum() - (P * W).sum() + c.reshape(())) / n_times_cout
    return loss, 2.0 * R / n_times_cout


def groupnorm_reads_twice(B, C, HW, groups):
    return HW >= 2048


def groupnorm_silu_fwd(x, gamma, beta, groups, eps, silu, force_slices=False, chan_part=None, pre_form=0):
    import torch.nn.functional as F
    cd = torch.float64 if x.dtype == torch.float64 else torch.float32
    xf = x.to(cd)
    B, C = x.shape[0], x.shape[1]
    g = xf.reshape(B, groups, -1)
    mean = g.mean(-1)
    var = g.var(-1, unbiased=False)
    if chan_part is not None:       # the producer's statistics are USED (a stale or foreign chan_part shows in the result)
        n = float(x.numel() // (B * groups))
        tot = chan_part.double().sum(1).reshape(B, groups, C // groups, 2).sum(2)          # (B, G, 2)
        mean = (tot[..., 0] / n).to(cd)
        var = (tot[..., 1] / n - (tot[..., 0] / n) ** 2).clamp_min(0).to(cd)
        shp = [B, groups] + [1] * (x.dim() - 1)
        xh = (xf.reshape(B, groups, C // groups, *x.shape[2:]) - mean.reshape(shp)) * torch.rsqrt(var + eps).reshape(shp)
        y = xh.reshape(x.shape) * gamma.to(cd).reshape([1, C] + [1] * (x.dim() - 2)) + beta.to(cd).reshape([1, C] + [1] * (x.dim() - 2))
    else:
        y = F.group_norm(xf, groups, gamma.to(cd), beta.to(cd), eps)
    if silu:
        y = F.silu(y)
    stats = torch.stack([mean.reshape(-1), torch.rsqrt(var + eps).reshape(-1)], 1).contiguous()
    return y.to(x.dtype), stats


def groupnorm_silu_bwd(dy, x, gamma, beta, stats, groups, silu, ds=None, force_slices=False):
    """Closed-form GroupNorm(+SiLU) input gradient from the saved (mean, rstd); checked against autograd in
    tests/test_host_cpu.py."""
    B, C = x.shape[0], x.shape[1]
    shape = [1, C] + [1] * (x.dim() - 2)
    mean = stats[:, 0].reshape(B, groups, 1)
    rstd = stats[:, 1].reshape(B, groups, 1)
    cd = torch.float64 if x.dtype == torch.float64 else torch.float32
    gamma, beta = gamma.to(cd), beta.to(cd)
    xh = ((x.to(cd).reshape(B, groups, -1) - mean.to(cd)) * rstd.to(cd)).reshape(x.shape)
    dz = dy.to(cd)
    if silu:
        z = xh * gamma.reshape(shape) + beta.reshape(shape)
        sig = torch.sigmoid(z)
        dz = dz * sig * (1 + z * (1 - sig))
    g = (dz * gamma.reshape(shape)).reshape(B, groups, -1)
    xg = xh.reshape(B, groups, -1)
    dx = rstd * (g - g.mean(-1, keepdim=True) - xg * (g * xg).mean(-1, keepdim=True))
    dx = dx.reshape(x.shape).to(x.dtype)
    if ds is not None:            # mos_groupnorm_silu_bwd_nhwc_res: the rounded norm gradient + the bypass gradient, rounded
        dx = (dx.float() + ds.float()).to(x.dtype)
    return dx


def layernorm_fwd(x, gamma, beta, eps, need_stats=True):
    xf = x.float()
    mean = xf.mean(-1, keepdim=True)
    var = ((xf - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    y = ((xf - mean) * rstd * gamma + beta).to(x.dtype)
    return y, (torch.cat([mean, rstd], -1) if need_stats else None)


def layernorm_bwd(dy, x, gamma, stats):
    """dx = rstd * (g - mean(g) - xhat * mean(g xhat)), g = dy * gamma (closed form the HIP kernel implements)."""
    mean, rstd = stats[:, :1].float(), stats[:, 1:].float()
    xh = (x.float() - mean) * rstd
    g = dy.float() * gamma
    return (rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))).to(x.dtype)


def add_layernorm_fwd(x, r, gamma, beta, eps, need_stats=True, half_dtype=None):
    """mos_add_layernorm_fwd: s = x + r rounded to the stream dtype of x, y = LN(s) in the half dtype."""
    if r is not None:
        s = (x.float() + r.float()).to(x.dtype)
        hd = r.dtype
    else:
        s = x
        hd = x.dtype if x.dtype != torch.float32 else (half_dtype or torch.float16)
    sf = s.float()
    mean = sf.mean(-1, keepdim=True)
    var = ((sf - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    y = ((sf - mean) * rstd * gamma + beta).to(hd)
    return s, y, (torch.cat([mean, rstd], -1) if need_stats else None)


def add_layernorm_bwd(dy, ds, s, gamma, stats, half_copy=False):
    """mos_add_layernorm_bwd: LN_bwd rounded to the stream dtype (as the unfused kernel does), + ds, rounded again."""
    mean, rstd = stats[:, :1].float(), stats[:, 1:].float()
    xh = (s.float() - mean) * rstd
    g = dy.float() * gamma
    dx = rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    if ds is not None:
        dx = dx.to(s.dtype).float() + ds.float()
    dx = dx.to(s.dtype)
    return dx, (dx.to(dy.dtype) if half_copy else None)


def geglu_fwd(h):
    a, g = h.float().chunk(2, dim=-1)
    return (a * torch.nn.functional.gelu(g)).to(h.dtype)


def geglu_bwd(dy, h):
    a, g = h.float().chunk(2, dim=-1)
    cdf = 0.5 * (1 + torch.erf(g * 0.7071067811865476))
    pdf = 0.3989422804014327 * torch.exp(-0.5 * g * g)
    d = dy.float()
    return torch.cat([d * g * cdf, d * a * (cdf + g * pdf)], -1).to(h.dtype)


def quick_gelu_fwd(x):
    z = x.float()
    return (z * torch.sigmoid(1.702 * z)).to(x.dtype)


def quick_gelu_bwd(dy, x):
    z = x.float()
    sg = torch.sigmoid(1.702 * z)
    return (dy.float() * sg * (1 + 1.702 * z * (1 - sg))).to(x.dtype)


def softmax_rows(x, scale, out=None):
    y = torch.softmax(x.float() * scale, -1).to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def single_head_attention_nograd(q, k, v, scale):
    """scores and probabilities are rounded to the half type between the GEMMs, like the kernel sequence (q pre-scaled
    by the power of two below `scale`, exact in half)."""
    import math
    p2 = 2.0 ** math.floor(math.log2(scale)) if scale > 0 else 1.0
    s = ((q * p2).float() @ k.float().transpose(-1, -2)).to(q.dtype)
    p = torch.softmax(s.float() * (scale / p2), -1).to(q.dtype)
    return (p.float() @ v.float()).to(q.dtype)


def conv3x3_nhwc(x, w_ohwi, bias=None, tbias=None, residual=None, upsample2x=False, split_k=True, gn_stats=False):
    """fp32 convolution of the half inputs, + bias + per-sample bias, rounded ONCE, then + residual rounded again.
    gn_stats: (y, chan_part) with ONE tile per image -- per-channel sum / sum of squares of the stored values."""
    if gn_stats:
        y = conv3x3_nhwc(x, w_ohwi, bias, tbias, residual, upsample2x, split_k)
        yf = y.float()
        part = torch.stack([yf.sum((2, 3)), (yf * yf).sum((2, 3))], dim=-1)[:, None].contiguous()      # (B, 1, C, 2)
        return y, part
    import torch.nn.functional as F
    xf = x.float()
    if upsample2x:
        xf = F.interpolate(xf, scale_factor=2.0, mode='nearest')
    y = F.conv2d(xf, w_ohwi.float().permute(0, 3, 1, 2), bias, padding=1)
    if tbias is not None:
        y = y + tbias.float()[:, :, None, None]
    y = y.to(x.dtype)
    if residual is not None:
        y = (y.float() + residual.float()).to(x.dtype)
    return y.contiguous(memory_format=torch.channels_last)


def conv3x3_s2_nhwc(x, w_ohwi, bias=None, pad_mode=1, split_k=True):
    """fp32 3x3 / stride-2 convolution of the half inputs, rounded once. pad_mode 1: padding 1; 2: F.pad(x, (0, 1, 0, 1)), padding 0."""
    import torch.nn.functional as F
    xf = x.float()
    if pad_mode == 2:
        xf = F.pad(xf, (0, 1, 0, 1))
    y = F.conv2d(xf, w_ohwi.float().permute(0, 3, 1, 2), bias, stride=2, padding=1 if pad_mode == 1 else 0)
    return y.to(x.dtype).contiguous(memory_format=torch.channels_last)
