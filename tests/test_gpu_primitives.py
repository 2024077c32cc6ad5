"""GPU parity of every HIP primitive (through the C-ABI via mixofshow.hip.ops) against the oracle's torch
emulation (oracle/emu_ops.py, fp32 math on the same half-precision inputs) at the SD-1.5 shapes.

Tolerances: outputs are rounded once to fp16/bf16; we allow a few ulps of the output dtype relative to the
largest magnitude in the tensor (fp16 eps = 9.8e-4, bf16 eps = 7.8e-3) — the 1e-3 fp16 tolerance of
BASELINE.json is checked end-to-end on denoised latents in test_gpu_end_to_end.py.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]


def _tol(dtype, ref, ulps=4.0):
    eps = 9.8e-4 if dtype == torch.float16 else 7.9e-3
    return ulps * eps * max(1.0, float(ref.abs().max()))


def _check(name, got, ref, dtype, ulps=4.0):
    got, ref = got.float(), ref.float()
    assert got.shape == ref.shape, f'{name}: shape {got.shape} vs {ref.shape}'
    assert torch.isfinite(got).all(), f'{name}: non-finite output'
    err = (got - ref).abs().max().item()
    tol = _tol(dtype, ref, ulps)
    print(f'[parity] {name}: max_abs_err={err:.3e} tol={tol:.3e} ref_max={ref.abs().max().item():.3e}')
    assert err <= tol, f'{name}: max abs err {err:.3e} > {tol:.3e}'


@pytest.fixture(scope='module')
def ops():
    import mixofshow.hip.ops as ops
    from mixofshow.hip import lib
    lib.load()
    return ops


@pytest.fixture(scope='module')
def emu():
    from oracle import emu_ops
    return emu_ops


def _lora_factors(sites, r, K, dev, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    downs = [(torch.randn(r, K, generator=g) * 0.05).to(dev) for _ in sites]
    ups = [(torch.randn(n, r, generator=g) * 0.05).to(dev) for n in sites]
    return downs, ups


@pytest.mark.parametrize('dtype', DTYPES)
def test_lora_pack_exact(ops, emu, dtype):
    dev = 'cuda'
    downs, ups = _lora_factors([320, 320, 320], 4, 320, dev, 0)
    got = ops.lora_pack(downs, ups, [1.0, 0.7, 0.3], 320, dtype, dev)
    ref = emu.lora_pack(downs, ups, [1.0, 0.7, 0.3], 320, dtype, dev)
    for g, r, n in zip(got, ref, ('A16', 'A16T', 'Bp16', 'BpT')):
        assert torch.equal(g, r), n


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K,sites', [
    (16384, 960, 320, [320, 320, 320]),   # L0 fused qkv, B=4
    (16384, 320, 320, [320]),             # L0 out-proj
    (308, 640, 768, [320, 320]),          # cross K/V on 4x77 text tokens (ragged M)
    (4096, 1920, 640, [640, 640, 640]),   # L1 fused qkv
    (1024, 1280, 1280, [1280]),           # L2
    (100, 320, 320, [320]),               # ragged tile
    (4928, 768, 768, [768]),              # CLIP q_proj on 64x77 tokens
])
def test_lora_linear_fwd_bwd(ops, emu, dtype, M, N, K, sites):
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev, dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, dtype)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    downs, ups = _lora_factors(sites, 4, K, dev, 2)
    alphas = [1.0] * len(sites)
    A16, A16T, Bp16, BpT = ops.lora_pack(downs, ups, alphas, K, dtype, dev)
    t = ops.lora_down(x, A16)
    t_ref = emu.lora_down(x, A16)
    _check('lora_down', t, t_ref, dtype)
    y = ops.linear_fwd(x, W, t_ref, Bp16, bias)
    y_ref = emu.linear_fwd(x, W, t_ref, Bp16, bias)
    _check(f'linear_fwd[{M}x{N}x{K}]', y, y_ref, dtype)
    y0 = ops.linear_fwd(x, W)
    _check('linear_fwd_plain', y0, emu.linear_fwd(x, W), dtype)
    # backward
    dy = torch.randn(M, N, generator=g).to(dev, dtype)
    Wt = W.t().contiguous()
    dx, dA, dB = ops.linear_bwd(dy, x, Wt, t_ref, A16T, BpT, lora_cols=4 * len(sites))
    dx_r, dA_r, dB_r = emu.linear_bwd(dy, x, Wt, t_ref, A16T, BpT)
    _check('linear_bwd.dx', dx, dx_r, dtype)
    # LoRA factor grads are fp32 sums over M tokens of half-precision products
    _check('linear_bwd.dA16', dA, dA_r, dtype, ulps=2.0)
    _check('linear_bwd.dBpT', dB, dB_r, dtype, ulps=2.0)
    dx2, _, _ = ops.linear_bwd(dy, x, Wt, None, None, None)
    _check('linear_bwd_plain.dx', dx2, emu.linear_bwd(dy, x, Wt, None, None, None)[0], dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K,sites', [
    (16384, 960, 320, [320, 320, 320]),   # L0 fused qkv, B=4            (128 x 64 tiles)
    (16384, 320, 320, [320]),             # L0 out-proj
    (308, 640, 768, [320, 320]),          # cross K/V on 4x77 text tokens (ragged M, 64 x 64 tiles)
    (4096, 1920, 640, [640, 640, 640]),   # L1 fused qkv                 (128 x 128 tiles)
    (4096, 640, 640, [640]),              # L1 out-proj                  (64 x 128 tiles)
    (1024, 1280, 1280, [1280]),           # L2                           (64 x 64 tiles)
    (256, 1280, 1280, [1280]),            # L3
    (100, 320, 320, [320]),               # ragged tile
    (4928, 2304, 768, [768, 768, 768]),   # CLIP fused q/k/v on 64x77 tokens
    (72, 328, 200, [328]),                # K % 64 != 0: the column-masked variant
    (512, 320, 64, [320]),                # one K tile: the DMA ring's prologue runs past the end of K
])
def test_lora_linear_fused_fwd_bwd(ops, emu, dtype, M, N, K, sites):
    """One-launch forward (down projection fused into the GEMM) and two-launch backward (dx+dt, then both factor
    gradients written straight into per-site fp32 targets, with and without accumulation) vs the emulation."""
    dev = 'cuda'
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(M, K, generator=g).to(dev, dtype)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev, dtype)
    bias = (torch.randn(N, generator=g) * 0.1).to(dev)
    downs, ups = _lora_factors(sites, 4, K, dev, 2)
    alphas = [1.0, 0.7, 0.3][:len(sites)]
    A16, A16T, Bp16, BpT = ops.lora_pack(downs, ups, alphas, K, dtype, dev)
    y, t = ops.linear_fused_fwd(x, W, A16, Bp16, bias)
    y_ref, t_ref = emu.linear_fused_fwd(x, W, A16, Bp16, bias)
    _check(f'fused_fwd.t[{M}x{K}]', t, t_ref, dtype)
    _check(f'fused_fwd.y[{M}x{N}x{K}]', y, y_ref, dtype)
    dy = torch.randn(M, N, generator=g).to(dev, dtype)
    Wt = W.t().contiguous()

    def targets(fill, acc):
        out = []
        for d, u, a in zip(downs, ups, alphas):
            out.append((torch.full_like(d, fill), torch.full_like(u, fill), a, u.shape[0], acc, acc))
        return out

    for fill, acc in ((float('nan'), False), (0.25, True)):      # overwrite must ignore what is there; accumulate adds
        tg, tr = targets(fill, acc), targets(fill, acc)
        dx = ops.linear_fused_bwd(dy, x, Wt, t_ref, A16T, BpT, tg, 4)
        dx_r = emu.linear_fused_bwd(dy, x, Wt, t_ref, A16T, BpT, tr, 4)
        _check('fused_bwd.dx', dx, dx_r, dtype)
        for i, (a, b) in enumerate(zip(tg, tr)):
            _check(f'fused_bwd.down_grad[{i}] acc={acc}', a[0], b[0], dtype, ulps=2.0)
            _check(f'fused_bwd.up_grad[{i}] acc={acc}', a[1], b[1], dtype, ulps=2.0)
    # determinism of the in-kernel ordered reduction: bit-identical on a second launch
    t1, t2 = targets(0.0, False), targets(0.0, False)
    ops.linear_fused_bwd(dy, x, Wt, t_ref, A16T, BpT, t1, 4)
    ops.linear_fused_bwd(dy, x, Wt, t_ref, A16T, BpT, t2, 4)
    for a, b in zip(t1, t2):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # frozen factors: only dx (and dt) are produced; one site's targets missing
    dx3 = ops.linear_fused_bwd(dy, x, Wt, t_ref, A16T, BpT, None, 4)
    _check('fused_bwd.dx (frozen LoRA)', dx3, dx_r, dtype)
    part = targets(0.0, False)
    part[0] = (None, part[0][1], part[0][2], part[0][3], False, False)
    ops.linear_fused_bwd(dy, x, Wt, t_ref, A16T, BpT, part, 4, need_dx=False)
    _check('fused_bwd.up_grad (no dx, down frozen)', part[0][1], tr[0][1] - 0.25, dtype, ulps=2.0)


@pytest.mark.parametrize('dtype', DTYPES)
def test_lora_pack_all_equals_per_group_pack(ops, emu, dtype):
    """mos_lora_pack_all (descriptor table in device memory, all groups in one launch) == mos_lora_pack per group;
    the registry repacks after an in-place parameter update and only then."""
    from mixofshow.hip import functional as F
    dev = torch.device('cuda', 0)
    specs = [([320, 320, 320], 320), ([320], 320), ([640, 640], 768), ([1280], 1280), ([768], 768)]
    reg = F.LoraPackRegistry(dev, dtype)
    groups = []
    for i, (sites, K) in enumerate(specs):
        downs, ups = _lora_factors(sites, 4, K, dev, 10 + i)
        groups.append((downs, ups, [1.0, 0.5, 0.25][:len(sites)], K))
    for downs, ups, alphas, K in groups:
        reg.get(downs, ups, alphas, K)
    for downs, ups, alphas, K in groups:
        got = reg.get(downs, ups, alphas, K)
        ref = ops.lora_pack(downs, ups, alphas, K, dtype, dev)
        for a, b, n in zip(got, ref, ('A16', 'A16T', 'Bp16', 'BpT')):
            assert torch.equal(a, b), n
    e0 = reg.epoch
    reg.get(*groups[0])
    assert reg.epoch == e0                                  # nothing changed: no repack
    groups[3][0][0].mul_(2.0)                               # optimiser-style in-place update of one down factor
    got = reg.get(*groups[3])
    assert reg.epoch == e0 + 1
    assert torch.equal(got[0], ops.lora_pack(*groups[3][:3], groups[3][3], dtype, dev)[0])
    assert torch.equal(reg.get(*groups[1])[2], ops.lora_pack(*groups[1][:3], groups[1][3], dtype, dev)[2])


def _qkv(B, Nq, Nkv, C, dtype, seed, fused):
    g = torch.Generator(device='cpu').manual_seed(seed)
    if fused and Nq == Nkv:
        buf = (torch.randn(B, Nq, 3 * C, generator=g)).to('cuda', dtype)
        return buf[..., :C], buf[..., C:2 * C], buf[..., 2 * C:]
    q = torch.randn(B, Nq, C, generator=g).to('cuda', dtype)
    kv = torch.randn(B, Nkv, 2 * C, generator=g).to('cuda', dtype)
    return q, kv[..., :C], kv[..., C:]


ATTN_CASES = [
    # B, H, Nq, Nkv, d
    (2, 8, 4096, 4096, 40),
    (2, 8, 1024, 1024, 80),
    (2, 8, 256, 256, 160),
    (2, 8, 64, 64, 160),
    (1, 8, 96, 96, 160),      # 8x12 map of 512x768: ragged kv tile
    (1, 8, 1536, 1536, 80),
    (2, 8, 4096, 77, 40),     # cross attention
    (2, 8, 1024, 77, 80),
    (2, 8, 256, 77, 160),
    (1, 8, 70, 77, 40),       # ragged queries
    (4, 8, 4096, 4096, 40),   # the bench shape (level-0 self attention, batch 4)
    (8, 8, 2000, 777, 40),    # large grid, ragged query and key tails, several key blocks
    (8, 8, 2048, 77, 40),     # batch 8 with exported probability columns
    (2, 8, 512, 600, 40),     # exported probability columns spread over several key blocks (keys 5, 70 / 599, 0)
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,Nq,Nkv,d', ATTN_CASES)
def test_attention_fwd_bwd(ops, emu, dtype, B, H, Nq, Nkv, d):
    C = H * d
    q, k, v = _qkv(B, Nq, Nkv, C, dtype, 3, fused=True)
    scale = d**-0.5
    cross = Nkv in (77, 600)
    tok = None
    if cross:
        rows = [[5, 70], [Nkv - 1, 0]] * ((B + 1) // 2)
        tok = torch.tensor(rows[:B], dtype=torch.int32, device='cuda').contiguous()
    o, lse, pcols = ops.attn_fwd(q, k, v, H, scale, tok_idx=tok)
    o_r, lse_r, pcols_r = emu.attn_fwd(q, k, v, H, scale, tok_idx=tok)
    _check(f'attn_fwd.o[{Nq}x{Nkv}x{d}]', o, o_r, dtype)
    _check('attn_fwd.lse', lse, lse_r, torch.float16, ulps=2.0)
    if cross:
        _check('attn_fwd.pcols', pcols, pcols_r, torch.float16, ulps=1.0)
    g = torch.Generator(device='cpu').manual_seed(4)
    dO = torch.randn(B, Nq, C, generator=g).to('cuda', dtype)
    dpc = (torch.randn(B, H, Nq, 2, generator=g) * 3.0).to('cuda') if cross else None
    dq, dk, dv = torch.empty_like(q.contiguous()), torch.empty_like(k.contiguous()), torch.empty_like(v.contiguous())
    ops.attn_bwd(q, k, v, o_r, lse_r, dO, H, scale, dq, dk, dv, tok_idx=tok, pcols=pcols_r, dpcols=dpc)
    dq_r, dk_r, dv_r = torch.empty_like(dq), torch.empty_like(dk), torch.empty_like(dv)
    emu.attn_bwd(q, k, v, o_r, lse_r, dO, H, scale, dq_r, dk_r, dv_r, tok_idx=tok, pcols=pcols_r, dpcols=dpc)
    _check('attn_bwd.dq', dq, dq_r, dtype, ulps=6.0)
    _check('attn_bwd.dk', dk, dk_r, dtype, ulps=6.0)
    _check('attn_bwd.dv', dv, dv_r, dtype, ulps=6.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,Nq,Nkv,d', [
    (2, 8, 4096, 77, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (2, 8, 64, 77, 160),     # the four UNet levels, CFG pair
    (2, 8, 6144, 77, 40),                   # 512x768 sample
    (1, 8, 70, 77, 40), (3, 8, 33, 5, 80),  # ragged queries, few keys, a probability tile that is not 16-byte aligned
    (1, 8, 130, 96, 160), (1, 8, 64, 3, 40),
])
def test_materialised_probabilities_split(ops, emu, dtype, B, H, Nq, Nkv, d):
    """mos_attn_probs / mos_attn_pv (the controller boundary with the full (B*H, N, 77) map, reference edlora.py:81-83) against the
    fp32 emulation, and their composition against the fused kernel; an in-place edit of the conditional half between the two
    (the reference's eval-mode rule, ptp_util.py:45-46) must reach the output."""
    C = H * d
    q, k, v = _qkv(B, Nq, Nkv, C, dtype, 17, fused=False)
    scale = d**-0.5
    P = ops.attn_probs(q, k, H, scale)
    P_r = emu.attn_probs(q, k, H, scale)
    assert P.shape == (B * H, Nq, Nkv) and P.dtype == dtype and P.is_contiguous()
    _check(f'attn_probs[{B}x{H}x{Nq}x{Nkv}x{d}]', P, P_r, dtype, ulps=2.0)
    rows = P.float().sum(-1)
    assert (rows - 1).abs().max().item() < (2e-2 if dtype == torch.bfloat16 else 3e-3)
    o = ops.attn_pv(P, v, H)
    _check('attn_pv on the kernel probabilities vs emulation', o, emu.attn_pv(P, v, H), dtype, ulps=2.0)
    o_f, _, _ = ops.attn_fwd(q, k, v, H, scale)
    _check('attn_probs -> attn_pv vs the fused kernel', o, o_f, dtype, ulps=6.0)
    if B >= 2:
        half = P.shape[0] // 2
        P2 = P.clone()
        e = P2[half:].float()
        e[:, :, :2] *= 3.0
        P2[half:] = (e / e.sum(-1, keepdim=True)).to(dtype)
        o2 = ops.attn_pv(P2, v, H)
        _check('attn_pv on edited probabilities', o2, emu.attn_pv(P2, v, H), dtype, ulps=2.0)
        nb = B // 2
        assert torch.equal(o2[:nb], o[:nb]) and not torch.equal(o2[nb:], o[nb:])


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,Nq,Nkv,d', [
    (2, 8, 4096, 77, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (2, 8, 64, 77, 160),     # the four UNet levels
    (1, 8, 70, 77, 40), (3, 8, 33, 5, 80), (1, 8, 130, 96, 160), (1, 8, 64, 3, 40),           # ragged queries, few / max keys
])
def test_materialised_probabilities_backward(ops, emu, dtype, B, H, Nq, Nkv, d):
    """mos_attn_pv_bwd / mos_attn_probs_bwd (round 6: the full-map controller boundary under autograd, reference edlora.py:81-83
    with a training-time store) against (a) the per-primitive emulation and (b) torch's fp32 autograd of
    softmax(scale q k^T) -> P' = edit(P) -> P' v with an extra loss that reads the map directly, as cal_attn_reg does; q / k / v
    and dq / dk / dv as strided slices of fused projection buffers; the key-side sums are deterministic (two runs bit-equal)."""
    from mixofshow.hip import functional as F_hip
    C = H * d
    g = torch.Generator(device='cpu').manual_seed(23)
    q = torch.randn(B, Nq, C, generator=g).to('cuda', dtype)
    kv = torch.randn(B, Nkv, 2 * C, generator=g).to('cuda', dtype)
    k, v = kv[..., :C], kv[..., C:]                      # column slices of ONE fused projection output
    dO = torch.randn(B, Nq, C, generator=g).to('cuda', dtype)
    w_map = (torch.randn(B * H, Nq, Nkv, generator=g) * 0.3).to('cuda')          # d(loss)/dP read straight off the stored map
    scale = d**-0.5
    P = ops.attn_probs(q, k, H, scale)
    dP, dv = ops.attn_pv_bwd(P, v, dO, H)
    dP_r, dv_r = emu.attn_pv_bwd(P, v, dO, H)
    _check(f'attn_pv_bwd.dP[{B}x{H}x{Nq}x{Nkv}x{d}]', dP, dP_r, dtype, ulps=2.0)
    _check('attn_pv_bwd.dv', dv, dv_r, dtype, ulps=4.0)
    dP2, dv2 = ops.attn_pv_bwd(P, v, dO, H)
    assert torch.equal(dP, dP2) and torch.equal(dv, dv2)
    dtot = (dP.float() + w_map).to(dtype)
    dq, dk = ops.attn_probs_bwd(q, k, P, dtot, H, scale)
    dq_r, dk_r = emu.attn_probs_bwd(q, k, P, dtot, H, scale)
    _check('attn_probs_bwd.dq', dq, dq_r, dtype, ulps=4.0)
    _check('attn_probs_bwd.dk', dk, dk_r, dtype, ulps=6.0)
    dq2, dk2 = ops.attn_probs_bwd(q, k, P, dtot, H, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2)
    # the autograd pair against torch's own fp32 autograd of the same computation
    qa, kva = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    Pa = F_hip.attn_probs(qa, kva[..., :C], H, scale)
    assert Pa.requires_grad
    oa = F_hip.attn_pv(Pa, kva[..., C:], H)
    ((oa.float() * dO.float()).sum() + (Pa.float() * w_map).sum()).backward()
    qf, kvf = q.float().requires_grad_(True), kv.float().requires_grad_(True)

    def hb(t):
        return t.reshape(B, t.shape[1], H, d).permute(0, 2, 1, 3)
    Pf = torch.softmax(hb(qf) @ hb(kvf[..., :C]).transpose(-1, -2) * scale, -1)
    of = (Pf @ hb(kvf[..., C:])).permute(0, 2, 1, 3).reshape(B, Nq, C)
    ((of * dO.float()).sum() + (Pf.reshape(B * H, Nq, Nkv) * w_map).sum()).backward()
    for name, a, b in (('dq', qa.grad, qf.grad), ('dkv', kva.grad, kvf.grad)):
        rel = ((a.float() - b).norm() / b.norm()).item()
        print(f'[parity] full-map autograd {name} [{B}x{H}x{Nq}x{Nkv}x{d} {dtype}]: rel-L2 {rel:.3e}')
        assert rel < (3e-2 if dtype == torch.bfloat16 else 5e-3), (name, rel)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,Nq,Nkv', [(2, 6144, 6144), (1, 200, 256), (2, 1000, 1536)])
def test_attention_fwd_drifting_scores(ops, emu, dtype, B, Nq, Nkv):
    """The online softmax of attn_fwd_kernel (d = 40) on scores that DRIFT from key tile to key tile: a case in which every
    tile's maximum is far above the previous one (the running maximum moves and O is rescaled at every step) and one in which
    it falls (it never moves), against the fp32 emulation."""
    H, d = 8, 40
    C = H * d
    g = torch.Generator(device='cpu').manual_seed(31)
    q = torch.randn(B, Nq, C, generator=g)
    k = torch.randn(B, Nkv, C, generator=g)
    v = torch.randn(B, Nkv, C, generator=g)
    for name, ramp in (('plain', None), ('rising', 1.0), ('falling', -1.0)):
        kk = k.clone()
        if ramp is not None:          # q.k grows / falls by ~30 raw units per 64-key tile for the first queries
            kk += ramp * 0.75 * (torch.arange(Nkv).float() / 64.0).floor()[None, :, None] * q[:, :1, :] / (q[:, :1, :].pow(2).mean(-1, keepdim=True).sqrt())
        qd, kd, vd = (t.to('cuda', dtype) for t in (q, kk, v))
        o1, lse1, _ = ops.attn_fwd(qd, kd, vd, H, d**-0.5)
        o_r, lse_r, _ = emu.attn_fwd(qd, kd, vd, H, d**-0.5)
        _check(f'attn_fwd drifting[{name} {B}x{Nq}x{Nkv}].o vs emulation', o1, o_r, dtype)
        _check(f'attn_fwd drifting[{name}].lse vs emulation', lse1, lse_r, torch.float16, ulps=2.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,Nq,Nkv', [(4, 4096, 4096), (1, 1024, 1024), (2, 256, 200), (1, 6144, 6144)])
def test_attention_bwd_dkdv_slot_interleaved_kernel(ops, emu, dtype, B, Nq, Nkv):
    """d = 40 dK/dV: attn_bwd_dkdv_pipe_kernel (slot-interleaved MFMA / softmax VALU; whole 64-query tiles) and, for the ragged
    case, attn_bwd_dkdv_kernel, against the fp32 emulation; dQ rides along."""
    H, d = 8, 40
    C = H * d
    q, k, v = _qkv(B, Nq, Nkv, C, dtype, 11, fused=(Nq == Nkv))
    scale = d**-0.5
    o_r, lse_r, _ = emu.attn_fwd(q, k, v, H, scale)
    g = torch.Generator(device='cpu').manual_seed(12)
    dO = torch.randn(B, Nq, C, generator=g).to('cuda', dtype)
    dq, dk, dv = torch.empty_like(q.contiguous()), torch.empty_like(k.contiguous()), torch.empty_like(v.contiguous())
    ops.attn_bwd(q, k, v, o_r, lse_r, dO, H, scale, dq, dk, dv)
    dq_r, dk_r, dv_r = torch.empty_like(dq), torch.empty_like(dk), torch.empty_like(dv)
    emu.attn_bwd(q, k, v, o_r, lse_r, dO, H, scale, dq_r, dk_r, dv_r)
    _check(f'attn_bwd dK/dV [{B}x{Nq}x{Nkv}].dk vs emulation', dk, dk_r, dtype, ulps=6.0)
    _check('attn_bwd dK/dV .dv vs emulation', dv, dv_r, dtype, ulps=6.0)
    _check('attn_bwd .dq vs emulation', dq, dq_r, dtype, ulps=6.0)


def test_attention_softmax_rescale_branch(ops, emu):
    """Force the online-softmax rescale: one key far above the rest in a LATE kv tile (cdna guide 5.4 rule 26)."""
    B, H, N, d = 1, 8, 512, 40
    g = torch.Generator(device='cpu').manual_seed(5)
    q = torch.randn(B, N, H * d, generator=g)
    k = torch.randn(B, N, H * d, generator=g)
    v = torch.randn(B, N, H * d, generator=g)
    k[:, 300] = q[:, 17] * 4.0          # spikes q.k for query 17 (and correlates for others) in tile 4
    k[:, 450] = -q[:, 99] * 4.0
    q, k, v = (t.to('cuda', torch.float16) for t in (q, k, v))
    o, lse, _ = ops.attn_fwd(q, k, v, H, d**-0.5)
    o_r, lse_r, _ = emu.attn_fwd(q, k, v, H, d**-0.5)
    _check('attn_fwd.spike.o', o, o_r, torch.float16)
    _check('attn_fwd.spike.lse', lse, lse_r, torch.float16, ulps=2.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('fh,fw,d', [(64, 96, 40), (32, 48, 80), (16, 24, 160), (8, 12, 160)])
def test_region_attention(ops, emu, dtype, fh, fw, d):
    B, H = 2, 8
    C = H * d
    N = fh * fw
    g = torch.Generator(device='cpu').manual_seed(6)
    q = torch.randn(B, N, C, generator=g).to('cuda', dtype)
    S = 5
    kv = torch.randn(S, B, 77, 2 * C, generator=g).to('cuda', dtype)
    k_src, v_src = kv[..., :C], kv[..., C:]
    # regionally_sample.sh boxes at 512x768 (+ one overlapping box), rounded like region_rewrite
    px = [[2, 2, 512, 184], [7, 184, 512, 345], [1, 488, 512, 747], [100, 150, 400, 300]]
    boxes = [(math.ceil(b[0] / 512 * fh), math.ceil(b[1] / 768 * fw), math.floor(b[2] / 512 * fh),
              math.floor(b[3] / 768 * fw)) for b in px]
    o = ops.region_attn_fwd(q, k_src, v_src, H, d**-0.5, boxes, fh, fw)
    o_r = emu.region_attn_fwd(q, k_src, v_src, H, d**-0.5, boxes, fh, fw)
    _check(f'region_attn[{fh}x{fw}x{d}]', o, o_r, dtype)
    # known answers: no regions == plain cross attention with the context keys
    o0 = ops.region_attn_fwd(q, k_src[:1], v_src[:1], H, d**-0.5, [], fh, fw)
    base, _, _ = emu.attn_fwd(q, k_src[0], v_src[0], H, d**-0.5)
    _check('region_attn.empty', o0, base, dtype)
    # one region covering everything with the context keys == base
    both = torch.stack([kv[0], kv[0]])
    o1 = ops.region_attn_fwd(q, both[..., :C], both[..., C:], H, d**-0.5, [(0, 0, fh, fw)], fh, fw)
    _check('region_attn.full_cover', o1, base, dtype)
    # two identical overlapping regions == one region (the covering regions' attentions are summed and divided by their count)
    one = ops.region_attn_fwd(q, k_src[:2], v_src[:2], H, d**-0.5, boxes[:1], fh, fw)
    twin = torch.stack([kv[0], kv[1], kv[1]])
    two = ops.region_attn_fwd(q, twin[..., :C], twin[..., C:], H, d**-0.5, [boxes[0], boxes[0]], fh, fw)
    _check('region_attn.twin_regions', two, one.float(), dtype)


@pytest.mark.parametrize('fh,fw,d', [(64, 96, 40), (32, 48, 80), (16, 24, 160), (8, 12, 160), (128, 256, 40)])
def test_region_attention_random_and_degenerate_boxes(ops, emu, fh, fw, d):
    """region_rewrite's box rule on the DEVICE (reference pipeline_regionally_t2iadapter.py:34-41,60-83):
    fractional boxes rounded ceil / floor like the reference -- so some collapse to ZERO area (start > end after rounding: the
    reference's slice is empty and the region contributes nothing), some touch or run along the borders, some coincide, up to
    the kernel's source limit; plus the all-degenerate layout (== base attention). 128 x 256 = the shipped 1024x2048 example's
    level 0."""
    import random
    from mixofshow.hip.ops import MOS_MAX_SOURCES
    B, H = 2, 8
    C = H * d
    N = fh * fw
    dtype = torch.float16
    g = torch.Generator(device='cpu').manual_seed(60)
    q = torch.randn(B, N, C, generator=g).to('cuda', dtype)
    kv = torch.randn(MOS_MAX_SOURCES, B, 77, 2 * C, generator=g).to('cuda', dtype)
    rnd = random.Random(61)
    n_zero = n_border = 0
    layouts = 3 if N > 8192 else 8
    for it in range(layouts):
        R = rnd.randint(1, MOS_MAX_SOURCES - 1)
        fr = []
        for r in range(R):
            kind = rnd.choice(['thin', 'border', 'any', 'any', 'dup'])
            if kind == 'thin':                      # narrower than one feature cell, not on a cell boundary: ceil(start) > floor(end)
                c0 = (rnd.randint(0, fw - 1) + 0.3) / fw
                fr.append([rnd.random() * 0.5, c0, 0.5 + rnd.random() * 0.5, c0 + 0.4 / fw])
            elif kind == 'border':
                fr.append([0.0, 0.0, 1.0, rnd.random()] if rnd.random() < 0.5 else [rnd.random() * 0.9, rnd.random() * 0.9, 1.0, 1.0])
            elif kind == 'dup' and fr:
                fr.append(list(fr[-1]))
            else:
                a, b = sorted([rnd.random(), rnd.random()])
                c, e = sorted([rnd.random(), rnd.random()])
                fr.append([a, c, b, e])
        boxes = [(math.ceil(f[0] * fh), math.ceil(f[1] * fw), math.floor(f[2] * fh), math.floor(f[3] * fw)) for f in fr]
        n_zero += sum(1 for b in boxes if b[2] <= b[0] or b[3] <= b[1])
        n_border += sum(1 for b in boxes if b[0] == 0 or b[1] == 0 or b[2] == fh or b[3] == fw)
        k_src, v_src = kv[:R + 1, ..., :C], kv[:R + 1, ..., C:]
        o = ops.region_attn_fwd(q, k_src, v_src, H, d**-0.5, boxes, fh, fw)
        o_r = emu.region_attn_fwd(q, k_src, v_src, H, d**-0.5, boxes, fh, fw)
        _check(f'region_attn random boxes [{fh}x{fw}x{d}] layout {it}: {boxes}', o, o_r, dtype)
    assert n_zero >= 1 and n_border >= 2, (n_zero, n_border)
    dead = [(5, 7, 5, 9), (3, 4, 2, 8), (fh, 0, fh, fw)]            # zero height, negative height, empty strip at the border
    o = ops.region_attn_fwd(q, kv[:4, ..., :C], kv[:4, ..., C:], H, d**-0.5, dead, fh, fw)
    base, _, _ = emu.attn_fwd(q, kv[0, ..., :C], kv[0, ..., C:], H, d**-0.5)
    _check('region_attn: only zero-area regions == base attention', o, base, dtype)


@pytest.mark.parametrize('fh,fw,d,R', [(64, 96, 40, 12), (32, 48, 80, 20), (16, 24, 160, 9), (8, 12, 160, 17)])
def test_region_attention_more_regions_than_one_launch_holds(ops, emu, fh, fw, d, R):
    """VERDICT r05 missing #2: the reference loops over an unbounded region_list (pipeline_regionally_t2iadapter.py:60-83);
    one launch of the kernel holds MOS_MAX_SOURCES-1 = 8 boxes. Longer lists run in chunks that share the whole list's
    per-query count (mos_region_cross_attn_fwd_chunk): checked against the per-primitive emulation AND against the oracle's
    restatement of region_rewrite itself (oracle/region_ref.py, head-batched fp32), with overlaps across chunk boundaries,
    queries covered only by late chunks, uncovered queries (context prompt) and a zero-area box in the middle of the list."""
    import random
    from oracle import region_ref
    from mixofshow.hip.ops import MOS_MAX_SOURCES
    assert R > MOS_MAX_SOURCES - 1
    B, H = 2, 8
    C = H * d
    N = fh * fw
    dtype = torch.float16
    g = torch.Generator(device='cpu').manual_seed(70 + R)
    q = torch.randn(B, N, C, generator=g).to('cuda', dtype)
    kv = torch.randn(R + 1, B, 77, 2 * C, generator=g).to('cuda', dtype)
    rnd = random.Random(71 + R)
    fr = []
    for r in range(R):
        if r == R // 2:
            c0 = (rnd.randint(0, fw - 1) + 0.3) / fw
            fr.append([0.1, c0, 0.9, c0 + 0.4 / fw])              # collapses to zero area after ceil / floor
        elif r in (0, 1, R - 2):
            fr.append([0.1, 0.1, 0.45, 0.4 + 0.05 * (r % 3)])       # three boxes (first and last chunk) share one area
        elif r == R - 1:
            fr.append([0.55, 0.6, 1.0, 1.0])                        # a corner that (mostly) only the LAST chunk covers
        else:
            a, b = sorted([rnd.random() * 0.6, rnd.random() * 0.6])
            c, e = sorted([rnd.random() * 0.7, rnd.random() * 0.7])
            fr.append([a, c, max(b, a + 1.5 / fh), max(e, c + 1.5 / fw)])
    boxes = region_ref.region_boxes_ref(fr, fh, fw)
    cnt = torch.zeros(fh, fw)
    for h0, w0, h1, w1 in boxes:
        cnt[h0:h1, w0:w1] += 1
    assert (cnt == 0).any() and (cnt > 2).any() and any(b[2] <= b[0] or b[3] <= b[1] for b in boxes)
    k_src, v_src = kv[..., :C], kv[..., C:]
    o = ops.region_attn_fwd(q, k_src, v_src, H, d**-0.5, boxes, fh, fw)
    o_r = emu.region_attn_fwd(q, k_src, v_src, H, d**-0.5, boxes, fh, fw)
    _check(f'region_attn {R} regions [{fh}x{fw}x{d}] vs emulation', o, o_r, dtype, ulps=4.0 + R / 4)   # one half rounding per chunk
    # the oracle's region_rewrite on head-batched fp32 tensors (reference layout: (B*H, N, d))
    def hb(t):                                                      # (B, n, C) -> (B*H, n, d)
        return t.float().cpu().reshape(B, t.shape[1], H, d).permute(0, 2, 1, 3).reshape(B * H, t.shape[1], d)

    class _A:
        scale, upcast_attention, upcast_softmax = d**-0.5, False, False

    base, _, _ = emu.attn_fwd(q, k_src[0], v_src[0], H, d**-0.5)
    rl = [(hb(k_src[r + 1]), hb(v_src[r + 1]), fr[r]) for r in range(R)]
    ref = region_ref.region_rewrite_ref(_A, hb(base), hb(q), rl, fh * 8, fw * 8)
    ref = ref.reshape(B, H, N, d).permute(0, 2, 1, 3).reshape(B, N, C)
    _check(f'region_attn {R} regions [{fh}x{fw}x{d}] vs oracle region_rewrite', o.cpu(), ref, dtype, ulps=4.0 + R / 4)
    # chunking must not change a list that fits one launch: first 8 boxes through both entry points
    o8 = ops.region_attn_fwd(q, k_src[:9], v_src[:9], H, d**-0.5, boxes[:8], fh, fw)
    o8_r = emu.region_attn_fwd(q, k_src[:9], v_src[:9], H, d**-0.5, boxes[:8], fh, fw)
    _check('region_attn 8 regions (one launch)', o8, o8_r, dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n,cin,cout', [(20000, 320, 320), (3000, 768, 320), (5000, 1280, 1280), (84, 768, 640),
                                        (777, 320, 2560)])
def test_gram_and_lsq(ops, emu, dtype, n, cin, cout):
    g = torch.Generator(device='cpu').manual_seed(7)
    X = torch.randn(n, cin, generator=g).to('cuda', dtype)
    Y = torch.randn(n, cout, generator=g).to('cuda', dtype)

    def fresh():
        return (torch.zeros(cin, cin, dtype=torch.float64, device='cuda'),
                torch.zeros(cout, cin, dtype=torch.float64, device='cuda'),
                torch.zeros(1, dtype=torch.float64, device='cuda'))

    G, P, c = fresh()
    ops.gram_accumulate(X, Y, G, P, c)
    ops.gram_accumulate(X, Y, G, P, c)     # accumulation across calls (concepts)
    Gr, Pr, cr = fresh()
    emu.gram_accumulate(X, Y, Gr, Pr, cr)
    emu.gram_accumulate(X, Y, Gr, Pr, cr)
    for name, a, b in (('G', G, Gr), ('P', P, Pr), ('c', c, cr)):
        rel = ((a - b).abs().max() / b.abs().max()).item()
        print(f'[parity] gram.{name}: rel_err={rel:.3e}')
        assert rel < 2e-5, f'gram {name} rel err {rel}'
    W = (torch.randn(cout, cin, generator=g, dtype=torch.float64) * 0.05).cuda()
    loss, grad = ops.lsq_loss_grad(W, Gr, Pr, cr, 2.0 * n * cout)
    loss_r, grad_r = emu.lsq_loss_grad(W, Gr, Pr, cr, 2.0 * n * cout)
    assert abs(loss.item() - loss_r.item()) <= 1e-10 * abs(loss_r.item()) + 1e-14
    assert ((grad - grad_r).abs().max() / grad_r.abs().max()).item() < 1e-10
    # and the Gram-form loss equals the direct mean((X W^T - Y)^2) of the reference closure
    direct = ((torch.cat([X, X]).double() @ W.t() - torch.cat([Y, Y]).double())**2).mean()
    assert abs(loss.item() - direct.item()) <= 1e-8 * abs(direct.item())


def test_errors_are_loud(ops):
    x = torch.randn(64, 320)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.lora_down(x.half(), x.half())
    q = torch.randn(1, 64, 8 * 48, device='cuda', dtype=torch.float16)
    from mixofshow.hip.lib import MosHipError
    with pytest.raises(MosHipError, match='head dim'):
        ops.attn_fwd(q, q, q, 8, 1.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('silu', [True, False])
@pytest.mark.parametrize('B,C,H,W,G', [(4, 320, 64, 64, 32), (2, 1280, 8, 8, 32), (2, 128, 256, 256, 32),
                                       (2, 2560, 16, 16, 32), (1, 640, 32, 48, 32)])
def test_groupnorm_silu(ops, emu, dtype, silu, B, C, H, W, G):
    """Fused GroupNorm(+SiLU) vs torch fp32 group_norm/silu on the same half inputs (forward and input gradient)."""
    g = torch.Generator(device='cpu').manual_seed(8)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to('cuda', dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    y, stats = ops.groupnorm_silu_fwd(x, gamma, beta, G, 1e-5, silu)
    y_ref, stats_ref = emu.groupnorm_silu_fwd(x, gamma, beta, G, 1e-5, silu)
    _check(f'groupnorm.y[{B}x{C}x{H}x{W}]', y, y_ref, dtype, ulps=2.0)
    _check('groupnorm.stats', stats, stats_ref, torch.float16, ulps=0.05)
    dy = torch.randn(B, C, H, W, generator=g).to('cuda', dtype)
    dx = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats_ref, G, silu)
    xf = x.float().requires_grad_(True)
    out = torch.nn.functional.group_norm(xf, G, gamma, beta, 1e-5)
    out = torch.nn.functional.silu(out) if silu else out
    (dx_ref, ) = torch.autograd.grad(out, xf, dy.float())
    _check('groupnorm.dx', dx, dx_ref, dtype, ulps=3.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('silu', [True, False])
@pytest.mark.parametrize('B,C,H,W,G', [
    (4, 320, 64, 64, 32),      # level 0 (10 channels per group: 8-channel vectors straddle groups)
    (2, 640, 32, 32, 32),
    (2, 1280, 16, 16, 32),
    (2, 1280, 8, 8, 32),
    (2, 2560, 8, 8, 32),       # up-block concatenation: two vectors per thread
    (1, 1920, 16, 24, 32),     # 512x768 regional sample, skip concat
    (2, 960, 64, 64, 32),
    (1, 32, 8, 12, 8),         # tiny preset (4 channels per group)
    (4, 128, 128, 128, 32),    # VAE encoder stage (reduced)
])
def test_groupnorm_silu_channels_last(ops, emu, dtype, silu, B, C, H, W, G):
    """The NHWC kernels (channels_last tensors) vs torch fp32 group_norm/silu: forward, statistics, input gradient;
    the output keeps the channels_last format."""
    g = torch.Generator(device='cpu').manual_seed(8)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to('cuda', dtype).contiguous(memory_format=torch.channels_last)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    y, stats = ops.groupnorm_silu_fwd(x, gamma, beta, G, 1e-5, silu)
    assert y.stride() == x.stride()
    y_ref, stats_ref = emu.groupnorm_silu_fwd(x, gamma, beta, G, 1e-5, silu)
    _check(f'groupnorm_nhwc.y[{B}x{C}x{H}x{W}]', y, y_ref, dtype, ulps=2.0)
    _check('groupnorm_nhwc.stats', stats, stats_ref, torch.float16, ulps=0.05)
    dy = torch.randn(B, C, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=torch.channels_last)
    dx = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats_ref, G, silu)
    assert dx.stride() == x.stride()
    xf = x.float().contiguous().requires_grad_(True)
    out = torch.nn.functional.group_norm(xf, G, gamma, beta, 1e-5)
    out = torch.nn.functional.silu(out) if silu else out
    (dx_ref, ) = torch.autograd.grad(out, xf, dy.float().contiguous())
    _check('groupnorm_nhwc.dx', dx, dx_ref, dtype, ulps=3.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('M,N,K,lora', [
    (16384, 320, 1280, False),    # FF2 level 0, training batch (N = 320: 64-wide tiles)
    (12288, 320, 1280, False),    # FF2 level 0, 512x768 CFG pair
    (3072, 640, 2560, False), (768, 1280, 5120, False), (192, 1280, 5120, False), (1024, 1280, 5120, False),
    (12288, 320, 320, False),     # proj_out 1x1 + residual
    (4096, 640, 640, True),       # a LoRA site with a residual (where: Transformer2DModel)
    (300, 320, 328, False),       # ragged M, K % 64 != 0
])
def test_gemm_residual_epilogue(ops, emu, dtype, M, N, K, lora):
    """mos_lora_linear_fwd_ex(residual): bit-identical to the GEMM followed by torch's half add (same rounding points), and
    within tolerance of the fp32 emulation."""
    g = torch.Generator(device='cpu').manual_seed(12)
    x = torch.randn(M, K, generator=g).to('cuda', dtype)
    W = (torch.randn(N, K, generator=g) * K ** -0.5).to('cuda', dtype)
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    r = torch.randn(M, N, generator=g).to('cuda', dtype)
    A16 = Bp16 = None
    if lora:
        downs, ups = _lora_factors([N], 4, K, 'cuda', 3)
        A16, _, Bp16, _ = ops.lora_pack(downs, ups, [1.0], K, dtype, x.device)
    y, t = ops.linear_fwd_ex(x, W, A16, Bp16, b, residual=r, need_t=lora)
    if lora:
        y0, t0 = ops.linear_fused_fwd(x, W, A16, Bp16, b)
        assert torch.equal(t, t0)
    else:
        y0 = ops.linear_fwd(x, W, None, None, b)
    exact = torch.equal(y, y0 + r)
    print(f'[parity] gemm+residual[{M}x{N}x{K} lora={lora}] bit-identical to GEMM + add: {exact}')
    assert exact
    _check('gemm+residual vs emulation', y, emu.linear_fwd_ex(x, W, A16, Bp16, b, residual=r)[0], dtype, ulps=2.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('silu', [True, False])
@pytest.mark.parametrize('B,C,H,W', [
    (4, 640, 32, 32), (2, 640, 32, 48),      # 40-channel columns, two groups each (resident: 1024 / 1536 pixels)
    (4, 320, 32, 32),                        # four 10-channel groups per column
    (2, 1280, 16, 24), (4, 1280, 8, 8),      # one group per column
    (2, 2560, 8, 12), (4, 2560, 16, 16),     # 80-channel columns
    (2, 1920, 16, 24), (4, 960, 16, 16),     # 120-channel columns: 2 / 4 groups, 15 vectors per pixel
    (2, 1920, 32, 48), (2, 960, 32, 48),     # 120-channel columns too large for registers: slice kernels
    (4, 320, 64, 64), (2, 320, 64, 96),      # level 0: slice kernels either way
    (1, 640, 5, 8),                          # fewer vectors than one wave
])
def test_groupnorm_column_kernel(ops, emu, dtype, silu, B, C, H, W):
    """The one-launch column kernel (register-resident slabs; the library's choice wherever a slab fits) against the three-launch
    slice kernels (MOS_GN_FORCE_SLICES) and the fp32 emulation: forward, statistics, input gradient, gradient + bypass."""
    g = torch.Generator(device='cpu').manual_seed(8)
    cl = torch.channels_last
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to('cuda', dtype).contiguous(memory_format=cl)
    dy = torch.randn(B, C, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=cl)
    ds = torch.randn(B, C, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=cl)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    y_ref, stats_ref = emu.groupnorm_silu_fwd(x, gamma, beta, 32, 1e-5, silu)
    xf = x.float().contiguous().requires_grad_(True)
    out = torch.nn.functional.group_norm(xf, 32, gamma, beta, 1e-5)
    out = torch.nn.functional.silu(out) if silu else out
    (dx_ref, ) = torch.autograd.grad(out, xf, dy.float().contiguous())
    res = {}
    for mode, slices in (('slices', True), ('auto', False)):
        y, stats = ops.groupnorm_silu_fwd(x, gamma, beta, 32, 1e-5, silu, force_slices=slices)
        dx = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats_ref, 32, silu, force_slices=slices)
        dxs = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats_ref, 32, silu, ds=ds, force_slices=slices)
        assert y.stride() == x.stride() and dx.stride() == x.stride()
        _check(f'groupnorm[{mode}].y[{B}x{C}x{H}x{W}]', y, y_ref, dtype, ulps=2.0)
        _check(f'groupnorm[{mode}].stats', stats, stats_ref, torch.float16, ulps=0.05)
        _check(f'groupnorm[{mode}].dx', dx, dx_ref, dtype, ulps=3.0)
        _check(f'groupnorm[{mode}].dx+ds', dxs, dx.float() + ds.float(), dtype, ulps=1.0)
        res[mode] = (y, stats, dx)
    same = torch.equal(res['auto'][0], res['slices'][0]) and torch.equal(res['auto'][2], res['slices'][2])
    print(f'[parity] groupnorm library choice [{B}x{C}x{H}x{W} silu={silu}] bit-identical to the slice kernels: {same}')
    _check('groupnorm library choice vs slice kernels: y', res['auto'][0], res['slices'][0], dtype, ulps=1.0)
    _check('groupnorm library choice vs slice kernels: dx', res['auto'][2], res['slices'][2], dtype, ulps=1.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C', [(16384, 320), (4096, 640), (1024, 1280), (256, 1280), (4928, 768), (77, 768), (6144, 320)])
def test_layernorm(ops, emu, dtype, rows, C):
    """Fused LayerNorm (half in/out, fp32 statistics) vs torch fp32 layer_norm on the same half inputs, fwd + dx."""
    g = torch.Generator(device='cpu').manual_seed(9)
    x = (torch.randn(rows, C, generator=g) * 2.0 + 0.5).to('cuda', dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    y, stats = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    y_ref, stats_ref = emu.layernorm_fwd(x, gamma, beta, 1e-5)
    _check(f'layernorm.y[{rows}x{C}]', y, y_ref, dtype, ulps=2.0)
    _check('layernorm.stats', stats, stats_ref, torch.float16, ulps=0.05)
    dy = torch.randn(rows, C, generator=g).to('cuda', dtype)
    dx = ops.layernorm_bwd(dy, x, gamma, stats_ref)
    xf = x.float().requires_grad_(True)
    (dx_ref, ) = torch.autograd.grad(torch.nn.functional.layer_norm(xf, (C, ), gamma, beta, 1e-5), xf, dy.float())
    _check('layernorm.dx', dx, dx_ref, dtype, ulps=3.0)
    _check('layernorm.dx (emulation formula)', emu.layernorm_bwd(dy, x, gamma, stats_ref), dx_ref, dtype, ulps=3.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('silu', [True, False])
@pytest.mark.parametrize('B,C,H,W', [(2, 320, 64, 64), (4, 640, 32, 32), (2, 1280, 8, 8), (1, 2560, 8, 8), (3, 64, 5, 7)])
def test_groupnorm_bwd_with_bypass_gradient(ops, emu, dtype, silu, B, C, H, W):
    """mos_groupnorm_silu_bwd_nhwc_res: dx = round(GN_bwd(dy)) + ds must equal the plain backward kernel followed by torch's
    half add (autograd's accumulation) -- same rounding points; and both vs the emulation's closed form."""
    g = torch.Generator(device='cpu').manual_seed(41)
    cl = torch.channels_last
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to('cuda', dtype).contiguous(memory_format=cl)
    dy = torch.randn(B, C, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=cl)
    ds = torch.randn(B, C, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=cl)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    _, stats = ops.groupnorm_silu_fwd(x, gamma, beta, 32, 1e-5, silu)
    plain = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, 32, silu)
    fused = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, 32, silu, ds=ds)
    assert fused.stride() == x.stride()
    # run-to-run bit equality, also of the slice form (64 x 64 maps; its block sums were LDS float atomics until round 6)
    _, stats2 = ops.groupnorm_silu_fwd(x, gamma, beta, 32, 1e-5, silu)
    assert torch.equal(stats, stats2) and torch.equal(fused, ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, 32, silu, ds=ds))
    exact = torch.equal(fused, plain + ds)
    print(f'[parity] groupnorm_bwd_res[{B}x{C}x{H}x{W} silu={silu}] bit-identical to kernel + add: {exact}')
    if not exact:
        _check('groupnorm_bwd_res vs kernel + add', fused, plain + ds, dtype, ulps=1.0)
    _check('groupnorm_bwd_res vs emulation', fused, emu.groupnorm_silu_bwd(dy, x, gamma, beta, stats, 32, silu, ds=ds), dtype,
           ulps=4.0)
    # round 6: ds read IN PLACE from a channel slice of a wider channels_last tensor (the gradient autograd hands to one input of
    # a torch.cat along the channels) -- both halves of a concatenation's gradient, bit-identical to the dense read
    for lo, wide in ((0, C + 320), (640, C + 640)):
        big = torch.randn(B, wide, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=cl)
        sl = big[:, lo:lo + C]
        assert ops.nhwc_pixel_stride(sl) == wide and ops.nhwc_pixel_stride(ds) == C and not sl.is_contiguous(memory_format=cl)
        got = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, 32, silu, ds=sl)
        want = ops.groupnorm_silu_bwd(dy, x, gamma, beta, stats, 32, silu, ds=sl.contiguous(memory_format=cl))
        assert torch.equal(got, want), f'sliced bypass gradient (channels {lo}..{lo + C} of {wide}) differs from its dense copy'


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C', [(8192, 320), (2048, 640), (512, 1280), (2464, 768), (7, 24), (33, 2048)])
def test_add_layernorm_half_stream_bit_identical_to_add_then_layernorm(ops, emu, dtype, rows, C):
    """mos_add_layernorm_*: residual add fused into LayerNorm on a HALF stream (UNet blocks). s, y, stats and dx must equal
    the separate torch add + mos_layernorm_* kernels BIT FOR BIT (same rounding points), with and without r / ds."""
    g = torch.Generator(device='cpu').manual_seed(31)
    x = (torch.randn(rows, C, generator=g) * 2.0 + 0.5).to('cuda', dtype)
    r = torch.randn(rows, C, generator=g).to('cuda', dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    dy = torch.randn(rows, C, generator=g).to('cuda', dtype)
    ds = torch.randn(rows, C, generator=g).to('cuda', dtype)
    def same(name, a, b):
        # the two kernels evaluate the same expressions with the same rounding points; identical machine code is expected,
        # a differing FMA contraction by the compiler would show up as a last-place difference: report, and bound it
        exact = torch.equal(a, b)
        print(f'[parity] add_layernorm[{rows}x{C}] {name}: bit-identical to the unfused kernels: {exact}')
        if not exact:
            _check(f'add_layernorm.{name} vs unfused', a, b, dtype, ulps=1.0)

    s_ref = x + r
    y_ref, st_ref = ops.layernorm_fwd(s_ref, gamma, beta, 1e-5)
    s, y, st = ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5)
    assert torch.equal(s, s_ref)
    same('y', y, y_ref)
    _check('add_layernorm.stats', st, st_ref, torch.float16, ulps=0.01)
    s0, y0, st0 = ops.add_layernorm_fwd(s_ref, None, gamma, beta, 1e-5)
    assert s0 is s_ref
    same('y (no r)', y0, y_ref)
    _, y1, st1 = ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5, need_stats=False)
    assert st1 is None and torch.equal(y1, y)
    dx_ln = ops.layernorm_bwd(dy, s_ref, gamma, st_ref)
    dx, dxh = ops.add_layernorm_bwd(dy, None, s, gamma, st_ref)
    assert dxh is None
    same('dx', dx, dx_ln)
    dx2, _ = ops.add_layernorm_bwd(dy, ds, s, gamma, st_ref)
    assert torch.equal(dx2, dx + ds)                 # half(LN_bwd) + ds rounded once more: exactly autograd's accumulation
    e_s, e_y, e_st = emu.add_layernorm_fwd(x, r, gamma, beta, 1e-5)
    assert torch.equal(e_s, s)
    _check(f'add_layernorm.y[{rows}x{C}] vs emulation', y, e_y, dtype, ulps=2.0)
    _check('add_layernorm.dx vs emulation', dx2, emu.add_layernorm_bwd(dy, ds, s, gamma, st_ref)[0], dtype, ulps=3.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,C', [(2464, 768), (77, 768), (5, 64), (130, 2048)])
def test_add_layernorm_fp32_stream_vs_fp32_torch(ops, emu, dtype, rows, C):
    """fp32 residual stream + half branch (the CLIP tower under autocast): s = x32 + r, statistics / y / dx from the fp32
    sum, plus the half copy of dx for the branch -- vs torch fp32 autograd of the same expression."""
    g = torch.Generator(device='cpu').manual_seed(32)
    x = (torch.randn(rows, C, generator=g) * 2.0 + 0.5).cuda()
    r = torch.randn(rows, C, generator=g).to('cuda', dtype)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
    beta = (0.1 * torch.randn(C, generator=g)).cuda()
    dy = torch.randn(rows, C, generator=g).to('cuda', dtype)
    ds = torch.randn(rows, C, generator=g).cuda()
    s, y, st = ops.add_layernorm_fwd(x, r, gamma, beta, 1e-5)
    assert s.dtype == torch.float32 and y.dtype == dtype and torch.equal(s, x + r.float())
    sf = s.clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(sf, (C, ), gamma, beta, 1e-5)
    _check(f'add_layernorm32.y[{rows}x{C}]', y, y_ref.detach(), dtype, ulps=1.0)
    (dln_ref, ) = torch.autograd.grad(y_ref, sf, dy.float())
    dx, dxh = ops.add_layernorm_bwd(dy, ds, s, gamma, st, half_copy=True)
    assert dx.dtype == torch.float32 and dxh.dtype == dtype
    ref = dln_ref + ds
    assert ((dx - ref).abs().max() <= 2e-5 * ref.abs().max() + 1e-5).item(), (dx - ref).abs().max().item()
    assert torch.equal(dxh, dx.to(dtype))
    dx0, dxh0 = ops.add_layernorm_bwd(dy, None, s, gamma, st, half_copy=True)
    assert ((dx0 - dln_ref).abs().max() <= 2e-5 * dln_ref.abs().max() + 1e-5).item() and torch.equal(dxh0, dx0.to(dtype))
    # stream without a pending branch (first CLIP layer: the embeddings themselves), half dtype named by the caller
    s1, y1, _ = ops.add_layernorm_fwd(x, None, gamma, beta, 1e-5, half_dtype=dtype)
    assert s1 is x and y1.dtype == dtype
    _check('add_layernorm32.y (no branch)', y1, torch.nn.functional.layer_norm(x, (C, ), gamma, beta, 1e-5), dtype, ulps=1.0)
    e_dx, e_dxh = emu.add_layernorm_bwd(dy, ds, s, gamma, st, half_copy=True)
    assert ((dx - e_dx).abs().max() <= 2e-5 * ref.abs().max() + 1e-5).item()


def test_lora_gradient_finals_deferred_equal_immediate():
    """TrainEngine's scope batches the LoRA factor gradients of a backward pass: the ordered final sums of all groups in ONE
    launch (mos_lora_grad_final_all) and -- round 5 -- the token reductions in one launch per padded-rank class
    (mos_lora_grad_all): bit-identical `.grad`s to the per-group launches, also on a second pass (persistent workspaces) and
    with accumulation into existing gradients; half and bfloat16 groups in one scope; several rank classes (4, 8, 12)."""
    from mixofshow.hip import functional as F_hip
    torch.manual_seed(3)
    dev = 'cuda'
    layers = []
    for (K, N, n_sites, dt) in ((320, 960, 3, torch.float16), (768, 2304, 3, torch.float16), (640, 640, 1, torch.float16),
                                (768, 1280, 2, torch.float16), (320, 320, 1, torch.float16), (320, 960, 3, torch.bfloat16),
                                (1280, 1280, 1, torch.bfloat16)):
        W = (torch.randn(N, K, device=dev) / K**0.5).to(dt)
        sites = []
        for _ in range(n_sites):
            down = torch.nn.Parameter(torch.randn(4, K, device=dev) * 0.05)
            up = torch.nn.Parameter(torch.randn(N // n_sites, 4, device=dev) * 0.05)
            sites.append((down, up, 0.7))
        layers.append((W, W.t().contiguous(), sites))
    xs = [torch.randn(2, 300 + 7 * i, W.shape[1], device=dev).to(W.dtype).requires_grad_(True) for i, (W, _, _) in enumerate(layers)]

    def run(mode, passes):
        for _, _, sites in layers:
            for d, u, _ in sites:
                d.grad, u.grad = torch.full_like(d, 0.25), torch.full_like(u, -0.5)     # accumulate on top of these
        store = F_hip.new_deferred_finals()
        store.defer_reduction = (mode == 'all')
        for _ in range(passes):
            scope = F_hip.direct_grad_accumulation(defer_finals=mode != 'immediate', store=store if mode != 'immediate' else None)
            with scope:
                loss = 0
                for (W, Wt, sites), x in zip(layers, xs):
                    loss = loss + F_hip.lora_linear(x, W, Wt, None, sites).float().square().mean()
                loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for _, _, sites in layers for d, u, _ in sites for p in (d, u)]

    for passes in (1, 3):             # 3: both pinned staging slots of the eager job-table upload, and the unchanged-image skip
        a, b, c = run('immediate', passes), run('finals', passes), run('all', passes)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), f'deferred final sums differ after {passes} pass(es)'
        assert all(torch.equal(x, y) for x, y in zip(a, c)), f'batched token reductions differ after {passes} pass(es)'
        assert all(torch.isfinite(x).all() and (x != 0.25).any() for x in a)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(64, 77, 3072), (3, 5, 24), (1, 8)])
def test_quick_gelu(ops, emu, dtype, shape):
    """CLIP text tower MLP activation x * sigmoid(1.702 x): forward and backward vs fp32 torch autograd (+ the emulation)."""
    g = torch.Generator(device='cpu').manual_seed(21)
    x = (torch.randn(*shape, generator=g) * 2.0).to('cuda', dtype)
    dy = torch.randn(*shape, generator=g).to('cuda', dtype)
    xr = x.float().requires_grad_(True)
    y_ref = xr * torch.sigmoid(1.702 * xr)
    (dx_ref, ) = torch.autograd.grad(y_ref, xr, dy.float())
    _check(f'quick_gelu.y{list(shape)}', ops.quick_gelu_fwd(x), y_ref.detach(), dtype, ulps=1.0)
    _check('quick_gelu.dx', ops.quick_gelu_bwd(dy, x), dx_ref, dtype, ulps=1.0)
    _check('quick_gelu.dx (emulation formula)', emu.quick_gelu_bwd(dy, x), dx_ref, dtype, ulps=1.0)
    from mixofshow.hip import functional as F_hip
    xa = x.clone().requires_grad_(True)
    F_hip.quick_gelu(xa).backward(dy)
    _check('quick_gelu autograd', xa.grad, dx_ref, dtype, ulps=1.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('rows,F', [(16384, 1280), (4096, 2560), (1024, 5120), (256, 5120), (100, 1280)])
def test_geglu(ops, emu, dtype, rows, F):
    """value * gelu(gate) (exact erf GELU, diffusers GEGLU) and its backward vs torch fp32 autograd."""
    g = torch.Generator(device='cpu').manual_seed(10)
    h = (torch.randn(rows, 2 * F, generator=g) * 1.5).to('cuda', dtype)
    y = ops.geglu_fwd(h)
    hf = h.float().requires_grad_(True)
    a, gate = hf.chunk(2, dim=-1)
    y_ref = a * torch.nn.functional.gelu(gate)
    _check(f'geglu.y[{rows}x{F}]', y, y_ref.detach(), dtype, ulps=2.0)
    dy = torch.randn(rows, F, generator=g).to('cuda', dtype)
    dh = ops.geglu_bwd(dy, h)
    (dh_ref, ) = torch.autograd.grad(y_ref, hf, dy.float())
    _check('geglu.dh', dh, dh_ref, dtype, ulps=2.0)
    _check('geglu.dh (emulation formula)', emu.geglu_bwd(dy, h), dh_ref, dtype, ulps=2.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,H,N,d', [(64, 12, 77, 64), (3, 12, 77, 64), (2, 12, 200, 64), (2, 8, 130, 40)])
def test_causal_attention_head_dim_64(ops, emu, dtype, B, H, N, d):
    """CLIP text tower: 12 heads x d = 64, 77 tokens, causal (key index > query index masked), forward + backward vs the
    emulation; q/k/v are column slices of one fused projection buffer, like the product path."""
    g = torch.Generator(device='cpu').manual_seed(12)
    C = H * d
    buf = torch.randn(B, N, 3 * C, generator=g).to('cuda', dtype)
    q, k, v = buf[..., :C], buf[..., C:2 * C], buf[..., 2 * C:]
    scale = d**-0.5
    o, lse, _ = ops.attn_fwd(q, k, v, H, scale, causal=True)
    o_r, lse_r, _ = emu.attn_fwd(q, k, v, H, scale, causal=True)
    _check(f'causal attn.o[{B}x{H}x{N}x{d}]', o, o_r, dtype)
    _check('causal attn.lse', lse, lse_r, torch.float16, ulps=0.5)
    # first token attends to itself only: its output is exactly v[0]
    _check('causal attn.o[token 0] == v[0]', o[:, 0], v[:, 0], dtype, ulps=1.0)
    dO = torch.randn(B, N, C, generator=g).to('cuda', dtype)
    dbuf, dref = torch.empty_like(buf), torch.empty_like(buf)
    ops.attn_bwd(q, k, v, o_r, lse_r, dO, H, scale, dbuf[..., :C], dbuf[..., C:2 * C], dbuf[..., 2 * C:], causal=True)
    emu.attn_bwd(q, k, v, o_r, lse_r, dO, H, scale, dref[..., :C], dref[..., C:2 * C], dref[..., 2 * C:], causal=True)
    for name, sl in (('dq', slice(0, C)), ('dk', slice(C, 2 * C)), ('dv', slice(2 * C, 3 * C))):
        _check(f'causal attn.{name}', dbuf[..., sl], dref[..., sl], dtype)
    # and the non-causal path at d = 64 (a head dim the UNet never uses)
    o2, _, _ = ops.attn_fwd(q, k, v, H, scale)
    _check('attn d64 non-causal', o2, emu.attn_fwd(q, k, v, H, scale)[0], dtype)


@pytest.mark.parametrize('dtype', DTYPES)
def test_softmax_rows_and_vae_single_head_attention(ops, emu, dtype):
    """VAE mid-block attention (one head, d = 512) as scores GEMM -> row softmax -> values GEMM on the library kernels vs
    the emulation and vs exact fp32 attention."""
    g = torch.Generator(device='cpu').manual_seed(13)
    x = (torch.randn(300, 4096, generator=g) * 3).to('cuda', dtype)
    _check('softmax_rows', ops.softmax_rows(x, 0.7), emu.softmax_rows(x, 0.7), dtype, ulps=2.0)
    x2 = (torch.randn(50, 1000, generator=g) * 3).to('cuda', dtype)
    _check('softmax_rows (ragged chunk count)', ops.softmax_rows(x2, 1.3), emu.softmax_rows(x2, 1.3), dtype, ulps=2.0)
    for rows, N in ((40, 16384), (24, 32768), (8, 20000)):         # 8 / 16 vectors per thread (1024 x 2048 images: 32768 keys)
        x3 = (torch.randn(rows, N, generator=g) * 3).to('cuda', dtype)
        _check(f'softmax_rows N={N}', ops.softmax_rows(x3, 0.9), emu.softmax_rows(x3, 0.9), dtype, ulps=2.0)
    for B, N, d in ((2, 1024, 512), (1, 8200, 512)):               # the second: two blocks of query rows
        q, k, v = ((torch.randn(B, N, d, generator=g) * 0.5).to('cuda', dtype) for _ in range(3))
        o = ops.single_head_attention_nograd(q, k, v, d**-0.5)
        _check(f'vae attention N={N} vs emulation', o, emu.single_head_attention_nograd(q, k, v, d**-0.5), dtype)
        exact = torch.softmax(q.float() @ k.float().transpose(-1, -2) * d**-0.5, -1) @ v.float()
        _check(f'vae attention N={N} vs exact fp32', o, exact, dtype, ulps=6.0)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,Cin,Cout,H,W,extras', [
    (4, 320, 320, 64, 64, 'tr'),      # level-0 ResNet convolutions: 8 x 16 x 64 halo tiles, 5 channel tiles
    (2, 320, 320, 64, 96, 'r'),       # 512x768 sample
    (2, 640, 640, 32, 48, 'u'),       # up-sampler into level 0 (64 x 96 output)
    (1, 128, 128, 512, 500, 't'),     # VAE stage: 16 x 16 x 128 tiles on 32-channel chunks, ragged
    (1, 512, 512, 130, 100, 'r'),     # ragged in both directions, 8 x 16 x 64 tiles
    (2, 960, 320, 64, 64, ''),        # last up block (Cin != Cout)
    (2, 640, 640, 32, 48, 'tr'),      # level 1 of a 512x768 sample: the map the one-launch column kernel would take (cpg 20)
    (4, 1280, 640, 32, 32, 't'),      # level 1, training batch, Cin != Cout
    (1, 256, 256, 128, 120, 'r'),     # VAE stage, cpg 8
    (2, 1920, 960, 32, 48, ''),       # cpg 30: a 120-channel range per workgroup
    (1, 128, 128, 512, 496, 'r'),     # VAE 512-px stage: 992 tiles per image -> the two-launch form of the consuming norm
])
def test_conv3x3_leaves_groupnorm_statistics(ops, emu, dtype, B, Cin, Cout, H, W, extras):
    """Round 6: mos_conv3x3_nhwc_gn -- the same convolution, bit for bit, plus per-(tile, channel) sum and sum
    of squares of the values it stored; mos_groupnorm_silu_fwd_nhwc_pre -- GroupNorm(+SiLU) from those sums (finalize over tiles
    + apply) against the three-launch form that re-reads the map, and against the fp32 emulation."""
    g = torch.Generator(device='cpu').manual_seed(35)
    x = torch.randn(B, Cin, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to('cuda', dtype)
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    up = 'u' in extras
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    tb = torch.randn(B, Cout, generator=g).to('cuda', dtype) if 't' in extras else None
    res = (torch.randn(B, Cout, Ho, Wo, generator=g) * 2).to('cuda', dtype).contiguous(memory_format=torch.channels_last) \
        if 'r' in extras else None
    w_fwd = w.permute(0, 2, 3, 1).contiguous()
    y_plain = ops.conv3x3_nhwc(x, w_fwd, bias, tb, res, up)
    y, part = ops.conv3x3_nhwc(x, w_fwd, bias, tb, res, up, gn_stats=True)
    assert part is not None and part.shape[0] == B and part.shape[2:] == (Cout, 2) and torch.equal(y, y_plain)
    yf = y.double()
    tot = part.double().sum(1)                                        # (B, Cout, 2)
    ref0, ref1 = yf.sum((2, 3)), (yf * yf).sum((2, 3))
    e0 = ((tot[..., 0] - ref0).abs().max() / ref1.sqrt().max()).item()
    e1 = ((tot[..., 1] - ref1).abs().max() / ref1.max()).item()
    print(f'[parity] conv3x3 GroupNorm statistics [{B}x{Cin}->{Cout}x{Ho}x{Wo} {extras}]: tiles {part.shape[1]}, sum err {e0:.2e}, sum-of-squares rel err {e1:.2e}')
    assert e0 < 1e-4 and e1 < 1e-5
    gamma = (torch.rand(Cout, generator=g) + 0.5).cuda()
    beta = (torch.randn(Cout, generator=g) * 0.2).cuda()
    for silu, eps in ((True, 1e-5), (False, 1e-6)):
        z_pre, st_pre = ops.groupnorm_silu_fwd(y, gamma, beta, 32, eps, silu, chan_part=part)
        z_ref, st_ref = ops.groupnorm_silu_fwd(y, gamma, beta, 32, eps, silu)
        z_emu, st_emu = emu.groupnorm_silu_fwd(y, gamma, beta, 32, eps, silu)
        _check(f'groupnorm from the convolution statistics vs emulation (silu={silu})', z_pre, z_emu, dtype, ulps=2.0)
        _check('groupnorm from the convolution statistics vs the re-reading form', z_pre, z_ref, dtype, ulps=1.0)
        assert (st_pre - st_emu).abs().max().item() <= 2e-5 * max(1.0, st_emu.abs().max().item())
        # the two forms of the _pre entry (one launch: every workgroup re-adds its groups' tile sums; two launches: a finalize
        # launch + the streaming launch -- the library picks by the prologue's cost, 512 x 512 VAE maps take the second): the same
        # sums in the same order up to the last double addition, then the same arithmetic
        z1, s1 = ops.groupnorm_silu_fwd(y, gamma, beta, 32, eps, silu, chan_part=part, pre_form=8)
        z2, s2 = ops.groupnorm_silu_fwd(y, gamma, beta, 32, eps, silu, chan_part=part, pre_form=4)
        assert (s1 - s2).abs().max().item() <= 1e-6 * max(1.0, s1.abs().max().item())
        _check('groupnorm _pre: one launch vs two launches', z1, z2, dtype, ulps=1.0)
        assert torch.equal(z_pre, z1) or torch.equal(z_pre, z2)
        dy = torch.randn_like(y)
        _check('groupnorm backward from those stats', ops.groupnorm_silu_bwd(dy, y, gamma, beta, st_pre, 32, silu),
               ops.groupnorm_silu_bwd(dy, y, gamma, beta, st_ref, 32, silu), dtype, ulps=2.0)
    # maps narrower than a halo tile take the raster form, which keeps no statistics
    xs = x[:, :, :12, :12].contiguous(memory_format=torch.channels_last)
    ys, none = ops.conv3x3_nhwc(xs, w_fwd, bias, None, None, False, gn_stats=True)
    assert none is None and torch.equal(ys, ops.conv3x3_nhwc(xs, w_fwd, bias))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('pad_mode', [1, 2])
@pytest.mark.parametrize('B,Cin,Cout,Hin,Win', [
    (2, 320, 320, 64, 96),       # UNet down-sampler of a 512x768 sample, level 0 -> 1
    (4, 640, 640, 32, 32),       # level 1 -> 2, training batch
    (2, 1280, 1280, 16, 24),     # level 2 -> 3: 8 x 12 output, split-K form
    (1, 128, 128, 256, 256),     # VAE encoder stage (128 x 128 tiles)
    (2, 256, 256, 128, 120),     # VAE encoder stage 2
    (1, 512, 512, 64, 64),       # VAE encoder stage 3
    (1, 64, 72, 33, 47),         # odd sizes, Cout not a multiple of the tile
    (1, 64, 8, 2, 3),            # smallest map
])
def test_conv3x3_stride2_nhwc(ops, emu, dtype, pad_mode, B, Cin, Cout, Hin, Win):
    """The down-samplers' 3x3 / stride-2 convolution on the raster implicit-GEMM kernel (round 6, mos_conv3x3_s2_nhwc) against
    fp32 torch conv2d on the same half operands: padding 1 (UNet Downsample2D) and the VAE encoder's F.pad(x, (0, 1, 0, 1)) +
    padding 0, whose padded copy the kernel never builds."""
    g = torch.Generator(device='cpu').manual_seed(33)
    x = torch.randn(B, Cin, Hin, Win, generator=g).to('cuda', dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to('cuda', dtype)
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    w_fwd = w.permute(0, 2, 3, 1).contiguous()
    y = ops.conv3x3_s2_nhwc(x, w_fwd, bias, pad_mode=pad_mode)
    y_ref = emu.conv3x3_s2_nhwc(x, w_fwd, bias, pad_mode=pad_mode)
    assert y.shape == y_ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    _check(f'conv3x3 stride 2 pad_mode {pad_mode} [{B}x{Cin}->{Cout}x{Hin}x{Win}]', y, y_ref, dtype)
    from mixofshow.hip import lib as _lib
    if _lib.load().mos_conv3x3_nhwc_workspace_bytes(B, y.shape[2], y.shape[3], Cin, Cout) > 0:
        y1 = ops.conv3x3_s2_nhwc(x, w_fwd, bias, pad_mode=pad_mode, split_k=False)
        _check('conv3x3 stride 2: split-K vs unsplit', y, y1, dtype, ulps=1.0)
    # through the module path (diffusers Downsample2D semantics), no gradient needed for x
    from mixofshow.models.unet_2d_condition import Downsample2D
    if Cin == Cout:
        ds = Downsample2D(Cin, padding=1 if pad_mode == 1 else 0).to('cuda', dtype).requires_grad_(False)
        with torch.no_grad():
            ds.conv.weight.copy_(w)
            ds.conv.bias.copy_(bias)
            _check('Downsample2D forward (HIP path)', ds(x), y_ref, dtype)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('B,Cin,Cout,H,W,extras', [
    (4, 320, 320, 64, 64, 'tr'),      # level-0 ResNet conv1 (+temb) / conv2 (+residual)     halo form, 8 x 16 x 64 tiles
    (4, 640, 640, 32, 32, 't'),       # level 1                                              halo form
    (4, 1280, 1280, 16, 16, 'r'),     # level 2                                              split-K raster form
    (4, 1280, 1280, 8, 8, 'tr'),      # level 3: 64-row raster tiles span images
    (2, 2560, 1280, 8, 8, ''),        # up-block concat input
    (2, 960, 320, 64, 64, 't'),       # last up block
    (1, 1920, 640, 32, 48, 'r'),      # 512x768 regional sample (non-square map)
    (2, 1280, 1280, 16, 16, 'u'),     # Upsample2D: nearest 2x folded into the HALO fetch, output 32x32
    (2, 640, 640, 32, 48, 'u'),       # Upsample2D into level 0 of a 512x768 sample (64x96)
    (1, 320, 320, 13, 21, 'u'),       # upsampled read with ragged halo tiles (26 x 42 output)
    (2, 128, 128, 96, 80, 't'),       # VAE-like stage
    (1, 128, 256, 250, 203, 'r'),     # VAE stage, ragged last tile in both directions
    (1, 128, 128, 512, 500, ''),      # VAE 512-px stage: the 16 x 16 x 128 halo tile on 32-channel chunks (>= 512 of them)
    (1, 256, 128, 300, 490, 't'),     # ... ragged in both directions, 8 chunks of 32 channels
    (1, 512, 512, 128, 128, 'r'),     # VAE 512-channel stage: the same tile (256 of them)
    (1, 512, 512, 64, 96, 'u'),       # VAE decoder up-sampler of a 512x768 sample: that tile with the upsampled halo fetch
    (1, 512, 512, 130, 100, ''),      # ... ragged, below the tile-count rule: 8 x 16 x 64 tiles
    (1, 64, 8, 5, 7, 'tr'),           # tiny / odd sizes (raster form: narrower than a halo tile)
    (1, 64, 72, 9, 17, 'tr'),         # one row / one column past a halo tile, Cout not a multiple of the 64-wide tile
    (2, 1280, 1280, 16, 24, 'tr'),    # 512x768 sample, level 2: split-K form (240 tiles, 4 K ranges)
    (2, 2560, 1280, 8, 12, 't'),      # level 3 up block: split-K, 360 K tiles
    (2, 1280, 1280, 8, 12, 'u'),      # Upsample2D into the 16x24 level: split-K with the upsampling gather
    (2, 1280, 640, 16, 24, ''),       # Cout = 640
])
def test_conv3x3_nhwc(ops, emu, dtype, B, Cin, Cout, H, W, extras):
    """Implicit-GEMM 3x3 convolution (channels_last) vs fp32 torch conv2d on the same half operands: forward with the
    fused per-sample bias / residual / upsample, and the backward-data form (flipped, transposed weight)."""
    g = torch.Generator(device='cpu').manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g).to('cuda', dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)).to('cuda', dtype)
    bias = (torch.randn(Cout, generator=g) * 0.1).cuda()
    up = 'u' in extras
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    tb = torch.randn(B, Cout, generator=g).to('cuda', dtype) if 't' in extras else None
    res = torch.randn(B, Cout, Ho, Wo, generator=g).to('cuda', dtype).contiguous(memory_format=torch.channels_last) \
        if 'r' in extras else None
    w_fwd = w.permute(0, 2, 3, 1).contiguous()
    y = ops.conv3x3_nhwc(x, w_fwd, bias, tb, res, up)
    assert y.shape == (B, Cout, Ho, Wo) and y.is_contiguous(memory_format=torch.channels_last)
    y_ref = emu.conv3x3_nhwc(x, w_fwd, bias, tb, res, up)
    _check(f'conv3x3[{B}x{Cin}->{Cout}x{H}x{W} {extras}]', y, y_ref, dtype)
    if Cout % 64 != 0:
        return                                   # backward-data contracts over Cout: the kernel needs Cout % 64 == 0 there
    # backward-data: dx = conv(dy, flip(W)^T) == autograd of the fp32 convolution
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).to('cuda', dtype).contiguous(memory_format=torch.channels_last)
    w_bwd = w.flip(2, 3).permute(1, 2, 3, 0).contiguous()
    dx = ops.conv3x3_nhwc(dy, w_bwd)
    xf = torch.zeros(B, Cin, Ho, Wo, device='cuda', requires_grad=True)
    (dx_ref, ) = torch.autograd.grad(torch.nn.functional.conv2d(xf, w.float(), None, padding=1), xf, dy.float())
    _check('conv3x3 backward-data', dx, dx_ref, dtype)
    # round 6: the same gradient read IN PLACE from a channel slice of a wider channels_last tensor (a torch.cat's gradient): every
    # kernel form (halo / raster / split-K) takes the pixel stride; bit-identical to the dense read
    for lo, wide in ((0, Cout + 320), (128, Cout + 192)):
        big = torch.zeros(B, wide, Ho, Wo, dtype=dtype, device='cuda').contiguous(memory_format=torch.channels_last)
        big.normal_(generator=torch.Generator(device='cuda').manual_seed(5))
        big[:, lo:lo + Cout] = dy
        sl = big[:, lo:lo + Cout]
        assert ops.nhwc_pixel_stride(sl) == wide
        assert torch.equal(ops.conv3x3_nhwc(sl, w_bwd), dx), f'dX from channels {lo}..{lo + Cout} of {wide} differs from the dense read'
    from mixofshow.hip import lib as _lib
    if _lib.load().mos_conv3x3_nhwc_workspace_bytes(B, Ho, Wo, Cin, Cout) > 0:
        # this shape took the split-K form: the unsplit kernel (same entry point without a workspace) must agree to the
        # rounding of the fp32 summation order
        y1 = ops.conv3x3_nhwc(x, w_fwd, bias, tb, res, up, split_k=False)
        _check(f'conv3x3 split-K vs unsplit [{B}x{Cin}->{Cout}x{H}x{W} {extras}]', y, y1, dtype, ulps=1.0)
        print(f'[parity] conv3x3 split-K [{B}x{Cin}->{Cout}x{Ho}x{Wo}] bit-identical to the unsplit kernel: {torch.equal(y, y1)}')
