"""Primitive kernel-backed ops: torch tensors in, torch tensors out, arithmetic in libmos_hip.so.

Every function requires device (HIP) tensors and raises otherwise — there is no eager fallback.
Shapes follow the reference's token-major `(B, N, H*d)` activations; see include/mos_hip.h.
"""
import ctypes

import torch

from . import lib as _lib
from .lib import MOS_BF16, MOS_F16, MOS_LORA_PAD, MOS_MAX_PCOLS, MOS_MAX_SOURCES

_DT = {torch.float16: MOS_F16, torch.bfloat16: MOS_BF16}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f'mixofshow.hip: dtype {t.dtype} not supported by the HIP kernels (float16 / bfloat16 only)')


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('mixofshow.hip: kernel-backed ops need tensors on a HIP device (cuda:N); got '
                               f'{t.device}. There is no CPU fallback in the product path.')


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _rows(t):
    """2-D view requirements: last dim contiguous, row stride multiple of 8 elements, 16-B aligned base."""
    assert t.dim() == 2 and t.stride(1) == 1, f'need a row-major 2-D view, got strides {t.stride()}'
    return t.stride(0)


# ------------------------------------------------------------------------------------------------
# LoRA-augmented linear
# ------------------------------------------------------------------------------------------------
def lora_pack(downs, ups, alphas, K, dtype, device):
    """Pack fp32 master LoRA factors of up to 4 sites sharing one input into MFMA operands.

    downs[g]: (r, K) fp32, ups[g]: (n_g, r) fp32. Returns A16 (16,K), A16T (K,16), Bp16 (N,16), BpT (16,N)."""
    n_sites = len(downs)
    r = downs[0].shape[0]
    if r > MOS_LORA_PAD:
        raise ValueError(f'LoRA rank {r} is not supported by the fused HIP path: the rank dimension is one '
                         f'{MOS_LORA_PAD}-wide MFMA operand (rank <= {MOS_LORA_PAD}); see INTEGRATION.md')
    if not (1 <= n_sites <= 4 and n_sites * r <= MOS_LORA_PAD):
        raise ValueError(f'{n_sites} LoRA sites of rank {r} do not fit one packed rank-{MOS_LORA_PAD} operand; fuse '
                         'fewer projections per call (mixofshow.models.attention falls back to one GEMM per projection)')
    _dev(*downs, *ups)
    s = _lib.LoraSites()
    s.n_sites, s.rank, s.K = n_sites, r, K
    keep = []
    n0 = 0
    for g in range(n_sites):
        d = downs[g].detach().reshape(r, K)
        u = ups[g].detach().reshape(-1, r)
        if d.dtype != torch.float32 or not d.is_contiguous():
            d = d.float().contiguous()
        if u.dtype != torch.float32 or not u.is_contiguous():
            u = u.float().contiguous()
        keep += [d, u]
        s.down[g] = d.data_ptr()
        s.up[g] = u.data_ptr()
        s.alpha[g] = float(alphas[g])
        s.n_begin[g] = n0
        s.n_rows[g] = u.shape[0]
        n0 += u.shape[0]
    s.N = n0
    A16 = torch.empty((MOS_LORA_PAD, K), dtype=dtype, device=device)
    A16T = torch.empty((K, MOS_LORA_PAD), dtype=dtype, device=device)
    Bp16 = torch.empty((n0, MOS_LORA_PAD), dtype=dtype, device=device)
    BpT = torch.empty((MOS_LORA_PAD, n0), dtype=dtype, device=device)
    L = _lib.load()
    _lib.check(L.mos_lora_pack(ctypes.byref(s), _DT[dtype], _p(A16), _p(A16T), _p(Bp16), _p(BpT), _stream()),
               'mos_lora_pack')
    return A16, A16T, Bp16, BpT


def lora_group_desc(downs, ups, alphas, K, A16, A16T, Bp16, BpT):
    """ctypes descriptor (mos_lora_group) of one fused-projection group; `downs`/`ups` are the fp32 master tensors
    (they must stay alive and in place: the descriptor holds raw pointers)."""
    n_sites = len(downs)
    r = downs[0].shape[0]
    if r > MOS_LORA_PAD:
        raise ValueError(f'LoRA rank {r} is not supported by the fused HIP path: the rank dimension is one '
                         f'{MOS_LORA_PAD}-wide MFMA operand (rank <= {MOS_LORA_PAD}); see INTEGRATION.md')
    if not (1 <= n_sites <= 4 and n_sites * r <= MOS_LORA_PAD):
        raise ValueError(f'{n_sites} LoRA sites of rank {r} do not fit one packed rank-{MOS_LORA_PAD} operand')
    g = _lib.LoraGroup()
    g.s.n_sites, g.s.rank, g.s.K = n_sites, r, K
    n0 = 0
    for i in range(n_sites):
        d, u = downs[i], ups[i]
        assert d.dtype == torch.float32 and u.dtype == torch.float32 and d.is_contiguous() and u.is_contiguous()
        g.s.down[i], g.s.up[i] = d.data_ptr(), u.data_ptr()
        g.s.alpha[i] = float(alphas[i])
        g.s.n_begin[i] = n0
        g.s.n_rows[i] = u.shape[0]
        n0 += u.shape[0]
    g.s.N = n0
    g.A16, g.A16T, g.Bp16, g.BpT = A16.data_ptr(), A16T.data_ptr(), Bp16.data_ptr(), BpT.data_ptr()
    return g


def lora_pack_all(groups_dev, n_groups, max_elems, dtype):
    """Pack every registered LoRA group in one launch. groups_dev: uint8 device tensor holding the mos_lora_group array."""
    _dev(groups_dev)
    L = _lib.load()
    _lib.check(L.mos_lora_pack_all(_p(groups_dev), int(n_groups), int(max_elems), _DT[dtype], _stream()),
               'mos_lora_pack_all')


def linear_fused_fwd(x, W, A16, Bp16, bias=None, need_t=True):
    """y[M,N] = x . W^T + (x . A16^T) . Bp16^T (+ bias) in ONE kernel; returns (y, t[M,16] | None)."""
    _dev(x, W, A16, Bp16, bias)
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.dtype == x.dtype and A16.shape == (MOS_LORA_PAD, K) and Bp16.shape == (N, MOS_LORA_PAD)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    t = torch.empty((M, MOS_LORA_PAD), dtype=x.dtype, device=x.device) if need_t else None
    L = _lib.load()
    _lib.check(L.mos_lora_linear_fused_fwd(_p(x), _rows(x), _p(W), _rows(W), _p(A16), _p(Bp16), _p(bias), _p(y), _rows(y),
                                           _p(t), M, N, K, _dt(x), _stream()), 'mos_lora_linear_fused_fwd')
    return y, t


def linear_fused_bwd(dy, x, Wt, t, A16T, BpT, grad_targets, rank, need_dx=True, defer=None, defer_reduction=False):
    """Backward of linear_fused_fwd in three launches (dx+dt; token reduction of both factor gradients; ordered final sum). grad_targets: None (LoRA factors frozen) or a list with one
    (down_grad, up_grad, alpha, n_rows, accumulate_down, accumulate_up) per site — fp32 contiguous tensors shaped like the parameters (either
    may be None); the kernel writes / accumulates into them directly. Returns dx | None.
    defer: None, or a callable (ws_key, nbytes) -> persistent fp32 workspace tensor; then the final sum is NOT launched and
    the (LoraFinalRec, ws) pair is returned as the second value for lora_grad_final_all. defer_reduction (with defer): the token
    reduction is not launched either -- the second value is (LoraFinalRec, LoraGradJob, keep) with `keep` the tensors the job's
    pointers refer to (dt, x, t, dy): the caller holds them until it has issued lora_grad_all over all jobs."""
    _dev(dy, x, Wt, t, A16T, BpT)
    M, N = dy.shape
    K = x.shape[1]
    dev = dy.device
    dx = torch.empty((M, K), dtype=dy.dtype, device=dev) if need_dx else None
    dt = torch.empty((M, MOS_LORA_PAD), dtype=dy.dtype, device=dev)
    L = _lib.load()
    g = None
    ws = None
    cols = 0
    if grad_targets:
        g = _lib.LoraGradOut()
        g.n_sites = len(grad_targets)
        n0 = 0
        for i, (dg, ug, alpha, n_rows, acc_d, acc_u) in enumerate(grad_targets):
            for tg in (dg, ug):
                assert tg is None or (tg.dtype == torch.float32 and tg.is_contiguous() and tg.is_cuda)
            assert dg is None or dg.numel() == rank * K
            assert ug is None or ug.numel() == rank * int(n_rows)
            g.rank = int(rank)
            g.down_grad[i] = dg.data_ptr() if dg is not None else None
            g.up_grad[i] = ug.data_ptr() if ug is not None else None
            g.alpha[i] = float(alpha)
            g.n_begin[i], g.n_rows[i] = n0, int(n_rows)
            g.accumulate_down[i], g.accumulate_up[i] = int(bool(acc_d)), int(bool(acc_u))
            n0 += int(n_rows)
        assert n0 == N
        cols = g.n_sites * g.rank
        nws = (L.mos_lora_bwd_workspace_bytes(M, N, K) + 3) // 4
        if defer is not None:
            ws = defer((BpT.data_ptr(), M, N, K), nws)
            rec = _lib.LoraFinalRec()
            if defer_reduction:
                job = _lib.LoraGradJob()
                _lib.check(L.mos_lora_linear_fused_bwd_deferred_all(
                    _p(dy), _rows(dy), _p(x), _rows(x), _p(Wt), _rows(Wt) if Wt is not None else 0, _p(t), _p(A16T), _p(BpT), _p(dt),
                    _p(dx), _rows(dx) if dx is not None else 0, ctypes.byref(g), _p(ws), M, N, K, int(cols), _dt(dy), _stream(),
                    ctypes.byref(rec), ctypes.byref(job)), 'mos_lora_linear_fused_bwd_deferred_all')
                return dx, (rec, job, (dt, x, t, dy))
            _lib.check(L.mos_lora_linear_fused_bwd_deferred(
                _p(dy), _rows(dy), _p(x), _rows(x), _p(Wt), _rows(Wt) if Wt is not None else 0, _p(t), _p(A16T), _p(BpT), _p(dt),
                _p(dx), _rows(dx) if dx is not None else 0, ctypes.byref(g), _p(ws), M, N, K, int(cols), _dt(dy), _stream(),
                ctypes.byref(rec)), 'mos_lora_linear_fused_bwd_deferred')
            return dx, rec
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
    _lib.check(L.mos_lora_linear_fused_bwd(_p(dy), _rows(dy), _p(x), _rows(x), _p(Wt), _rows(Wt) if Wt is not None else 0,
                                           _p(t), _p(A16T), _p(BpT), _p(dt), _p(dx), _rows(dx) if dx is not None else 0,
                                           ctypes.byref(g) if g is not None else None, _p(ws), M, N, K, int(cols),
                                           _dt(dy), _stream()), 'mos_lora_linear_fused_bwd')
    return (dx, None) if defer is not None else dx


def lora_grad_all(jobs_dev, byte_offset, n_jobs, total_blocks, nj, dtype, flops, nbytes):
    """ONE launch: the token reductions of every deferred LoRA gradient group of one padded-rank class (`n_jobs` records starting
    `byte_offset` bytes into the uint8 device table `jobs_dev`)."""
    L = _lib.load()
    _lib.check(L.mos_lora_grad_all(ctypes.c_void_p(jobs_dev.data_ptr() + int(byte_offset)), int(n_jobs), int(total_blocks), int(nj),
                                   MOS_F16 if dtype == torch.float16 else MOS_BF16, float(flops), float(nbytes), _stream()),
               'mos_lora_grad_all')


def lora_grad_final_all(recs_dev, n_recs, total_blocks):
    """ONE launch: the ordered final sums of every deferred LoRA gradient group (records uploaded by the caller)."""
    L = _lib.load()
    _lib.check(L.mos_lora_grad_final_all(_p(recs_dev), int(n_recs), int(total_blocks), _stream()), 'mos_lora_grad_final_all')


def lora_down(x, A16):
    """t[M,16] = x[M,K] . A16^T"""
    _dev(x, A16)
    M, K = x.shape
    t = torch.empty((M, MOS_LORA_PAD), dtype=x.dtype, device=x.device)
    L = _lib.load()
    _lib.check(L.mos_lora_down(_p(x), _rows(x), _p(A16), _p(t), M, K, _dt(x), _stream()), 'mos_lora_down')
    return t


def linear_fwd(x, W, t=None, Bp16=None, bias=None, out=None):
    """y[M,N] = x . W^T (+ t . Bp16^T) (+ bias);  W (N,K) in x.dtype, bias fp32."""
    _dev(x, W, t, Bp16, bias)
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.dtype == x.dtype
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    y = out if out is not None else torch.empty((M, N), dtype=x.dtype, device=x.device)
    L = _lib.load()
    _lib.check(L.mos_lora_linear_fwd(_p(x), _rows(x), _p(W), _rows(W), _p(t), _p(Bp16), _p(bias), _p(y), _rows(y),
                                     M, N, K, _dt(x), _stream()), 'mos_lora_linear_fwd')
    return y


def linear_fwd_ex(x, W, A16=None, Bp16=None, bias=None, residual=None, need_t=False):
    """mos_lora_linear_fwd_ex: y = x . W^T (+ (x . A16^T) . Bp16^T) (+ bias) with the residual epilogue:
    residual (M, N): added to the rounded result (== GEMM followed by a half add, one launch). Returns (y, t | None)."""
    _dev(x, W, A16, Bp16, bias, residual)
    M, K = x.shape
    N = W.shape[0]
    assert W.shape[1] == K and W.dtype == x.dtype and (A16 is None) == (Bp16 is None)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    epi = _lib.GemmEpilogue()
    epi.residual, epi.ldr = None, 0
    if residual is not None:
        assert residual.shape == (M, N) and residual.dtype == x.dtype
        epi.residual, epi.ldr = residual.data_ptr(), _rows(residual)
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    t = torch.empty((M, MOS_LORA_PAD), dtype=x.dtype, device=x.device) if (need_t and A16 is not None) else None
    L = _lib.load()
    _lib.check(L.mos_lora_linear_fwd_ex(_p(x), _rows(x), _p(W), _rows(W), _p(A16), _p(Bp16), _p(bias), _p(y), _rows(y), _p(t),
                                        M, N, K, _dt(x), ctypes.byref(epi), _stream()), 'mos_lora_linear_fwd_ex')
    return y, t


def linear_bwd(dy, x, Wt, t, A16T, BpT, need_dx=True, need_lora=True, lora_cols=16):
    """Backward of linear_fwd. Returns (dx | None, dA16 (16,K) fp32 | None, dBpT (16,N) fp32 | None)."""
    _dev(dy, x, Wt, t, A16T, BpT)
    M, N = dy.shape
    lora = BpT is not None
    K = Wt.shape[0] if Wt is not None else x.shape[1]
    dev = dy.device
    dx = torch.empty((M, K), dtype=dy.dtype, device=dev) if need_dx else None
    dt = torch.empty((M, MOS_LORA_PAD), dtype=dy.dtype, device=dev) if lora else None
    dA16 = torch.empty((MOS_LORA_PAD, K), dtype=torch.float32, device=dev) if (lora and need_lora) else None
    dBpT = torch.empty((MOS_LORA_PAD, N), dtype=torch.float32, device=dev) if (lora and need_lora) else None
    L = _lib.load()
    ws = None
    if lora:
        ws = torch.empty((L.mos_lora_bwd_workspace_bytes(M, N, K) + 3) // 4, dtype=torch.float32, device=dev)
    _lib.check(L.mos_lora_linear_bwd(_p(dy), _rows(dy), _p(x), _rows(x) if x is not None else 0,
                                     _p(Wt), _rows(Wt) if Wt is not None else 0, _p(t), _p(A16T), _p(BpT),
                                     _p(dt), _p(dx), _rows(dx) if dx is not None else 0, _p(dA16), _p(dBpT), _p(ws),
                                     M, N, K, int(lora_cols), _dt(dy), _stream()), 'mos_lora_linear_bwd')
    return dx, dA16, dBpT


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _attn_view(t):
    """(B, N, C) tensor, last dim contiguous; returns (batch_stride, row_stride)."""
    assert t.dim() == 3 and t.stride(2) == 1, f'need (B, N, C) with contiguous channels, got strides {t.stride()}'
    return t.stride(0), t.stride(1)


def _shape(q, k, v, o, heads, scale, causal=False):
    B, Nq, C = q.shape
    s = _lib.AttnShape()
    s.B, s.H, s.Nq, s.Nkv, s.d = B, heads, Nq, k.shape[1], C // heads
    s.q_bs, s.q_rs = _attn_view(q)
    s.k_bs, s.k_rs = _attn_view(k)
    s.v_bs, s.v_rs = _attn_view(v)
    s.o_bs, s.o_rs = _attn_view(o)
    s.scale = float(scale)
    s.causal = int(bool(causal))
    return s


def attn_fwd(q, k, v, heads, scale, tok_idx=None, need_lse=True, causal=False):
    """softmax(scale q k^T) v per head. q (B,Nq,C), k/v (B,Nkv,C) views (may be slices of fused buffers).

    Returns (o (B,Nq,C), lse (B,H,Nq) fp32 | None, pcols (B,H,Nq,T) fp32 | None).
    tok_idx: int32 (B,T) key indices whose softmax probabilities are exported (T <= 4)."""
    _dev(q, k, v, tok_idx)
    B, Nq, C = q.shape
    o = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    s = _shape(q, k, v, o, heads, scale, causal)
    lse = torch.empty((B, heads, Nq), dtype=torch.float32, device=q.device) if need_lse else None
    T = 0
    pcols = None
    if tok_idx is not None:
        assert tok_idx.dtype == torch.int32 and tok_idx.is_contiguous() and tok_idx.shape[0] == B
        T = tok_idx.shape[1]
        assert 1 <= T <= MOS_MAX_PCOLS
        pcols = torch.empty((B, heads, Nq, T), dtype=torch.float32, device=q.device)
    L = _lib.load()
    _lib.check(L.mos_attn_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(tok_idx), T, _p(pcols), ctypes.byref(s),
                              _dt(q), _stream()), 'mos_attn_fwd')
    return o, lse, pcols


def attn_bwd(q, k, v, o, lse, dO, heads, scale, dq, dk, dv, tok_idx=None, pcols=None, dpcols=None, causal=False):
    """Backward of attn_fwd; dq/dk/dv are preallocated (B,N,C) views (e.g. slices of one fused buffer)."""
    _dev(q, k, v, o, lse, dO, dq, dk, dv, tok_idx, pcols, dpcols)
    s = _shape(q, k, v, o, heads, scale, causal)
    g = _lib.AttnGradStrides()
    g.do_bs, g.do_rs = _attn_view(dO)
    g.dq_bs, g.dq_rs = _attn_view(dq)
    g.dk_bs, g.dk_rs = _attn_view(dk)
    g.dv_bs, g.dv_rs = _attn_view(dv)
    T = tok_idx.shape[1] if tok_idx is not None else 0
    if dpcols is not None:
        assert dpcols.dtype == torch.float32 and dpcols.is_contiguous() and pcols is not None
    L = _lib.load()
    ws = torch.empty((L.mos_attn_bwd_workspace_bytes(ctypes.byref(s)) + 3) // 4, dtype=torch.float32, device=q.device)
    _lib.check(L.mos_attn_bwd(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(dO), _p(tok_idx), T, _p(pcols), _p(dpcols),
                              _p(dq), _p(dk), _p(dv), _p(ws), ctypes.byref(s), ctypes.byref(g), _dt(q), _stream()),
               'mos_attn_bwd')
    return dq, dk, dv


_region_count_cache = {}


def region_total_count(boxes, feat_h, feat_w, device):
    """The reference's `count` tensor (pipeline_regionally_t2iadapter.py:56,80) for a whole region list: per query, how many
    boxes cover it -- a (feat_h*feat_w,) uint8 device tensor, built on the host from the integer boxes (no device work, no
    sync) and kept per (boxes, size, device): the same handful of layouts recurs at every layer and step of a call."""
    key = (tuple(tuple(int(v) for v in b) for b in boxes), int(feat_h), int(feat_w), str(device))
    t = _region_count_cache.get(key)
    if t is None:
        cnt = torch.zeros((feat_h, feat_w), dtype=torch.int32)
        for h0, w0, h1, w1 in key[0]:
            if h1 > h0 and w1 > w0:
                cnt[max(h0, 0):h1, max(w0, 0):w1] += 1
        assert int(cnt.max()) <= 255, 'more than 255 overlapping regions on one query'
        if len(_region_count_cache) > 256:
            _region_count_cache.clear()
        t = _region_count_cache[key] = cnt.to(torch.uint8).reshape(-1).to(device)
    return t


def region_attn_fwd(q, k_src, v_src, heads, scale, boxes, feat_h, feat_w):
    """Regional mask-and-blend cross attention.

    q (B,Nq,C); k_src/v_src (S,B,Nkv,C) views with contiguous channels, source 0 = context prompt,
    sources 1.. = regions; boxes: list of (h0, w0, h1, w1) integer feature-cell boxes per region -- any number of them
    (the reference's region_list is unbounded): up to MOS_MAX_SOURCES-1 in one launch, longer lists in chunks that share the
    whole list's per-query count (mos_region_cross_attn_fwd_chunk)."""
    _dev(q, k_src, v_src)
    assert k_src.dim() == 4 and v_src.dim() == 4 and k_src.shape[0] == len(boxes) + 1
    B, Nq, C = q.shape
    o = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    s = _shape(q, k_src[0], v_src[0], o, heads, scale)
    assert k_src.stride(0) == v_src.stride(0), 'k_src and v_src must share the source stride'
    L = _lib.load()
    per = MOS_MAX_SOURCES - 1
    total = region_total_count(boxes, feat_h, feat_w, q.device) if len(boxes) > per else None
    for c0 in range(0, max(len(boxes), 1), per):
        chunk = boxes[c0:c0 + per]
        r = _lib.RegionDesc()
        r.n_regions, r.feat_h, r.feat_w = len(chunk), int(feat_h), int(feat_w)
        for i, b in enumerate(chunk):
            for j in range(4):
                r.box[i][j] = int(b[j])
        r.src_stride = k_src.stride(0)
        # source 0 of a later chunk is the source before its first region (never read there: accumulate skips the context)
        _lib.check(L.mos_region_cross_attn_fwd_chunk(_p(q), _p(k_src[c0]), _p(v_src[c0]), _p(o), ctypes.byref(s),
                                                     ctypes.byref(r), _p(total), int(c0 > 0), _dt(q), _stream()),
                   'mos_region_cross_attn_fwd')
    return o


def attn_probs(q, k, heads, scale):
    """The reference's `attn.get_attention_scores(query, key)` (edlora.py:81) on the token-major tensors: q (B,Nq,C), k (B,Nkv,C)
    views -> softmax(scale q k^T) per head as a dense (B*heads, Nq, Nkv) tensor of q's dtype (index b * heads + h, the
    reference's head_to_batch_dim order). Nkv <= 96. For controllers that need the full map (no autograd here: see
    mixofshow.hip.functional.attn_probs / attn_pv for the differentiable pair)."""
    _dev(q, k)
    B, Nq, C = q.shape
    probs = torch.empty((B * heads, Nq, k.shape[1]), dtype=q.dtype, device=q.device)
    s = _shape(q, k, k, q, heads, scale)
    _lib.check(_lib.load().mos_attn_probs(_p(q), _p(k), _p(probs), ctypes.byref(s), _dt(q), _stream()), 'mos_attn_probs')
    return probs


def attn_pv(probs, v, heads):
    """torch.bmm(attention_probs, value) + batch_to_head_dim (edlora.py:83-85): probs (B*heads, Nq, Nkv) dense, v (B,Nkv,C) view
    -> (B, Nq, C)."""
    _dev(probs, v)
    B, Nkv, C = v.shape
    assert probs.dim() == 3 and probs.shape[0] == B * heads and probs.shape[2] == Nkv and probs.dtype == v.dtype
    probs = probs if probs.is_contiguous() else probs.contiguous()
    Nq = probs.shape[1]
    o = torch.empty((B, Nq, C), dtype=v.dtype, device=v.device)
    s = _shape(o, v, v, o, heads, 1.0)
    _lib.check(_lib.load().mos_attn_pv(_p(probs), _p(v), _p(o), ctypes.byref(s), _dt(v), _stream()), 'mos_attn_pv')
    return o


def _probs_bwd_ws(L, s, device):
    return torch.empty((L.mos_attn_probs_bwd_workspace_bytes(ctypes.byref(s)) + 3) // 4, dtype=torch.float32, device=device)


def attn_pv_bwd(probs, v, dO, heads):
    """Backward of attn_pv: probs (B*heads, Nq, Nkv) as the controller returned them, v (B,Nkv,C) view, dO (B,Nq,C)
    -> (dprobs dense like probs, dv (B,Nkv,C))."""
    _dev(probs, v, dO)
    B, Nkv, C = v.shape
    Nq = probs.shape[1]
    assert probs.shape == (B * heads, Nq, Nkv) and dO.shape == (B, Nq, C) and probs.dtype == v.dtype == dO.dtype
    probs = probs if probs.is_contiguous() else probs.contiguous()
    dO = dO if dO.stride(-1) == 1 else dO.contiguous()
    dprobs = torch.empty_like(probs)
    dv = torch.empty((B, Nkv, C), dtype=v.dtype, device=v.device)
    s = _shape(dO, v, v, dO, heads, 1.0)
    g = _lib.AttnGradStrides()
    g.do_bs, g.do_rs = _attn_view(dO)
    g.dv_bs, g.dv_rs = _attn_view(dv)
    g.dq_bs = g.dq_rs = g.dk_bs = g.dk_rs = 0
    L = _lib.load()
    ws = _probs_bwd_ws(L, s, v.device)
    _lib.check(L.mos_attn_pv_bwd(_p(probs), _p(v), _p(dO), _p(dprobs), _p(dv), _p(ws), ctypes.byref(s), ctypes.byref(g),
                                 _dt(v), _stream()), 'mos_attn_pv_bwd')
    return dprobs, dv


def attn_probs_bwd(q, k, probs, dprobs, heads, scale):
    """Backward of attn_probs: q (B,Nq,C), k (B,Nkv,C) views, probs = what attn_probs returned, dprobs = the gradient that
    reached the map (through attn_pv AND from whatever the controller's loss read) -> (dq (B,Nq,C), dk (B,Nkv,C))."""
    _dev(q, k, probs, dprobs)
    B, Nq, C = q.shape
    Nkv = k.shape[1]
    assert probs.shape == dprobs.shape == (B * heads, Nq, Nkv) and probs.dtype == q.dtype
    probs = probs if probs.is_contiguous() else probs.contiguous()
    dprobs = dprobs.to(probs.dtype)
    dprobs = dprobs if dprobs.is_contiguous() else dprobs.contiguous()
    dq = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
    dk = torch.empty((B, Nkv, C), dtype=k.dtype, device=k.device)
    s = _shape(q, k, k, q, heads, scale)
    g = _lib.AttnGradStrides()
    g.dq_bs, g.dq_rs = _attn_view(dq)
    g.dk_bs, g.dk_rs = _attn_view(dk)
    g.do_bs = g.do_rs = g.dv_bs = g.dv_rs = 0
    L = _lib.load()
    ws = _probs_bwd_ws(L, s, q.device)
    _lib.check(L.mos_attn_probs_bwd(_p(q), _p(k), _p(probs), _p(dprobs), _p(dq), _p(dk), _p(ws), ctypes.byref(s),
                                    ctypes.byref(g), _dt(q), _stream()), 'mos_attn_probs_bwd')
    return dq, dk


# ------------------------------------------------------------------------------------------------
# gradient-fusion least squares (Gram form)
# ------------------------------------------------------------------------------------------------
def gram_accumulate(X, Y, G, P, c):
    """G += X^T X ; P += Y^T X ; c += sum(Y^2).  X (n,Cin), Y (n,Cout) half/bf16; G,P,c fp64 on device."""
    _dev(X, Y, G, P, c)
    n, Cin = X.shape
    Cout = Y.shape[1]
    assert Y.shape[0] == n and G.shape == (Cin, Cin) and P.shape == (Cout, Cin) and c.numel() == 1
    assert G.dtype == P.dtype == c.dtype == torch.float64 and G.is_contiguous() and P.is_contiguous()
    L = _lib.load()
    ws = torch.empty((L.mos_gram_workspace_bytes(n, Cin, Cout) + 3) // 4, dtype=torch.float32, device=X.device)
    _lib.check(L.mos_gram_accumulate(_p(X), _rows(X), _p(Y), _rows(Y), n, Cin, Cout, _dt(X), _p(G), _p(P), _p(c),
                                     _p(ws), _stream()), 'mos_gram_accumulate')


def lsq_loss_grad(W, G, P, c, n_times_cout):
    """loss (0-dim fp64) and grad (Cout,Cin fp64) of mean((X W^T - Y)^2) from the Gram statistics."""
    _dev(W, G, P, c)
    Cout, Cin = W.shape
    assert W.dtype == torch.float64 and W.is_contiguous()
    loss = torch.empty((), dtype=torch.float64, device=W.device)
    grad = torch.empty_like(W)
    L = _lib.load()
    ws = torch.empty((L.mos_lsq_workspace_bytes(Cout, Cin) + 7) // 8, dtype=torch.float64, device=W.device)
    _lib.check(L.mos_lsq_loss_grad_gram(_p(W), _p(G), _p(P), _p(c), float(n_times_cout), Cout, Cin, _p(loss), _p(grad),
                                        _p(ws), _stream()), 'mos_lsq_loss_grad_gram')
    return loss, grad


# ------------------------------------------------------------------------------------------------
# fused GroupNorm (+ SiLU) — caller-side plumbing kernel
# ------------------------------------------------------------------------------------------------
def _same_layout(a, b):
    """Same shape and the same strides on every dimension of extent > 1 (the stride of a size-1 dimension is arbitrary)."""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


def _is_nhwc(x):
    return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def nhwc_pixel_stride(t):
    """Pixel stride (elements) of a (B, C, H, W) tensor whose memory is a CHANNEL SLICE of a channels_last tensor -- what autograd
    hands to one input of a torch.cat along the channels (the UNet's skip concatenations) -- or of a dense channels_last tensor
    (stride C); None for any other layout. The kernels that take a pixel stride read such a gradient in place."""
    if t.dim() != 4:
        return None
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    ps = sw if W > 1 else (sh if H > 1 else (sb // max(H * W, 1) if B > 1 else C))
    if C > 1 and sc != 1:
        return None
    if ps < C or ps % 8 or (W > 1 and sw != ps) or (H > 1 and sh != W * ps) or (B > 1 and sb != H * W * ps):
        return None
    if t.data_ptr() % 16:
        return None
    return int(ps)


def groupnorm_reads_twice(B, C, HW, groups):
    """True where groupnorm_silu_fwd on a channels_last (B, C, ..) map takes the three-launch slice form (statistics pass + apply
    pass): there the producing convolution's channel statistics (conv3x3_nhwc(..., gn_stats=True)) save a pass over the map."""
    return bool(_lib.load().mos_groupnorm_nhwc_reads_twice(int(B), int(C), int(HW), int(groups)))


def groupnorm_silu_fwd(x, gamma, beta, groups, eps, silu, force_slices=False, chan_part=None, pre_form=0):
    """x (B, C, *spatial) half, contiguous (NCHW) or channels_last (NHWC); gamma/beta fp32.
    Returns (y like x, same memory format; stats (B*G, 2) fp32). force_slices (channels_last): the three-launch slice kernels
    also where the one-launch column kernel applies (MOS_GN_FORCE_SLICES; parity tests, A/B). chan_part (channels_last):
    (B, tiles, C, 2) fp32 per-(pixel tile, channel) sum / sum of squares of x as its producer left them -- the statistics pass
    over x is skipped (mos_groupnorm_silu_fwd_nhwc_pre; pre_form 4 / 8 = MOS_GN_PRE_TWO_LAUNCHES / MOS_GN_PRE_ONE_LAUNCH pin the
    form the library otherwise picks: tests, tools)."""
    assert pre_form in (0, 4, 8)
    _dev(x, gamma, beta, chan_part)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    stats = torch.empty((B * groups, 2), dtype=torch.float32, device=x.device)
    L = _lib.load()
    if _is_nhwc(x) and chan_part is not None:
        assert chan_part.dtype == torch.float32 and chan_part.is_contiguous() and chan_part.dim() == 4 \
            and chan_part.shape[0] == B and chan_part.shape[2] == C and chan_part.shape[3] == 2
        ws = torch.empty((L.mos_groupnorm_nhwc_workspace_bytes(B, C, HW, groups) + 3) // 4, dtype=torch.float32, device=x.device)
        _lib.check(L.mos_groupnorm_silu_fwd_nhwc_pre(_p(x), _p(chan_part), int(chan_part.shape[1]), _p(gamma), _p(beta), _p(y),
                                                     _p(stats), _p(ws), B, C, HW, groups, float(eps), int(bool(silu)) | pre_form,
                                                     _dt(x), _stream()), 'mos_groupnorm_silu_fwd_nhwc_pre')
        return y, stats
    if _is_nhwc(x):
        ws = torch.empty((L.mos_groupnorm_nhwc_workspace_bytes(B, C, HW, groups) + 3) // 4, dtype=torch.float32, device=x.device)
        _lib.check(L.mos_groupnorm_silu_fwd_nhwc(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(ws), B, C, HW, groups,
                                                 float(eps), int(bool(silu)) | (2 if force_slices else 0), _dt(x), _stream()), 'mos_groupnorm_silu_fwd_nhwc')
        return y, stats
    assert x.is_contiguous(), 'groupnorm_silu_fwd needs a contiguous (NCHW) or channels_last (NHWC) tensor'
    ws = torch.empty((L.mos_groupnorm_workspace_bytes(B, C, HW, groups) + 3) // 4, dtype=torch.float32, device=x.device)
    _lib.check(L.mos_groupnorm_silu_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(ws), B, C, HW, groups,
                                        float(eps), int(bool(silu)), _dt(x), _stream()), 'mos_groupnorm_silu_fwd')
    return y, stats


def groupnorm_silu_bwd(dy, x, gamma, beta, stats, groups, silu, ds=None, force_slices=False):
    """dy must have x's memory format. ds (channels_last x only): gradient of a skip path around the norm, same format and
    dtype as x; the result is round(GN_bwd(dy)) + ds in one kernel."""
    _dev(dy, x, gamma, beta, stats, ds)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    dx = torch.empty_like(x)
    L = _lib.load()
    if _is_nhwc(x):
        assert _same_layout(dy, x)
        ws = torch.empty((L.mos_groupnorm_nhwc_workspace_bytes(B, C, HW, groups) + 3) // 4, dtype=torch.float32, device=x.device)
        if ds is not None:
            ps = nhwc_pixel_stride(ds)     # C for a dense map; wider for a channel slice of a concatenation's gradient
            assert ds.shape == x.shape and ps is not None and ds.dtype == x.dtype
            _lib.check(L.mos_groupnorm_silu_bwd_nhwc_res_ps(_p(dy), _p(ds), ps, _p(x), _p(gamma), _p(beta), _p(stats), _p(dx),
                                                            _p(ws), B, C, HW, groups, int(bool(silu)) | (2 if force_slices else 0),
                                                            _dt(x), _stream()), 'mos_groupnorm_silu_bwd_nhwc_res_ps')
            return dx
        _lib.check(L.mos_groupnorm_silu_bwd_nhwc(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dx), _p(ws), B, C, HW,
                                                 groups, int(bool(silu)) | (2 if force_slices else 0), _dt(x), _stream()), 'mos_groupnorm_silu_bwd_nhwc')
        return dx
    assert x.is_contiguous() and dy.is_contiguous() and ds is None, 'the bypass gradient is fused for channels_last only'
    ws = torch.empty((L.mos_groupnorm_workspace_bytes(B, C, HW, groups) + 3) // 4, dtype=torch.float32, device=x.device)
    _lib.check(L.mos_groupnorm_silu_bwd(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dx), _p(ws), B, C, HW, groups,
                                        int(bool(silu)), _dt(x), _stream()), 'mos_groupnorm_silu_bwd')
    return dx


# ------------------------------------------------------------------------------------------------
# row-wise operators of the transformer blocks: LayerNorm, GEGLU
# ------------------------------------------------------------------------------------------------
def layernorm_fwd(x, gamma, beta, eps, need_stats=True):
    """x (rows, C) half contiguous; gamma/beta fp32. Returns (y, stats (rows, 2) fp32 | None)."""
    _dev(x, gamma, beta)
    rows, C = x.shape
    assert x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    y = torch.empty_like(x)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if need_stats else None
    L = _lib.load()
    _lib.check(L.mos_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), rows, C, float(eps), _dt(x), _stream()),
               'mos_layernorm_fwd')
    return y, stats


def layernorm_bwd(dy, x, gamma, stats):
    _dev(dy, x, gamma, stats)
    rows, C = x.shape
    assert x.is_contiguous() and dy.is_contiguous() and dy.dtype == x.dtype
    dx = torch.empty_like(x)
    L = _lib.load()
    _lib.check(L.mos_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(stats), _p(dx), rows, C, _dt(x), _stream()), 'mos_layernorm_bwd')
    return dx


def add_layernorm_fwd(x, r, gamma, beta, eps, need_stats=True, half_dtype=None):
    """s = x + r, y = LayerNorm(s) (mos_add_layernorm_fwd). x (rows, C) contiguous in the residual-stream dtype (half, or
    fp32 for a half branch on an fp32 stream); r (rows, C) half or None (then s is x itself). y is half: r's dtype, else
    x's, else `half_dtype` (fp32 stream without r). Returns (s, y, stats)."""
    _dev(x, r, gamma, beta)
    rows, C = x.shape
    stream32 = x.dtype == torch.float32
    assert x.is_contiguous() and gamma.dtype == torch.float32 and beta.dtype == torch.float32
    assert r is None or (r.is_contiguous() and r.shape == x.shape and r.dtype in (torch.float16, torch.bfloat16))
    assert stream32 or r is None or r.dtype == x.dtype
    hd = r.dtype if r is not None else (x.dtype if not stream32 else (half_dtype or torch.float16))
    y = torch.empty((rows, C), dtype=hd, device=x.device)
    s = torch.empty_like(x) if r is not None else x
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if need_stats else None
    L = _lib.load()
    _lib.check(L.mos_add_layernorm_fwd(_p(x), _p(r), _p(gamma), _p(beta), _p(s) if r is not None else None, _p(y), _p(stats),
                                       rows, C, float(eps), _dt(y), int(stream32), _stream()), 'mos_add_layernorm_fwd')
    return s, y, stats


def add_layernorm_bwd(dy, ds, s, gamma, stats, half_copy=False):
    """dx = LN_bwd(dy) + ds in the stream dtype of s (ds None: no bypass gradient); half_copy (fp32 stream only): also the
    half rounding of dx for the branch input. Returns (dx, dx_half | None)."""
    _dev(dy, ds, s, gamma, stats)
    rows, C = s.shape
    stream32 = s.dtype == torch.float32
    assert s.is_contiguous() and dy.is_contiguous() and dy.shape == s.shape and dy.dtype in (torch.float16, torch.bfloat16)
    assert stream32 or dy.dtype == s.dtype
    assert ds is None or (ds.is_contiguous() and ds.dtype == s.dtype and ds.shape == s.shape)
    assert not half_copy or stream32
    dx = torch.empty_like(s)
    dxh = torch.empty_like(dy) if half_copy else None
    L = _lib.load()
    _lib.check(L.mos_add_layernorm_bwd(_p(dy), _p(ds), _p(s), _p(gamma), _p(stats), _p(dx), _p(dxh), rows, C, _dt(dy),
                                       int(stream32), _stream()), 'mos_add_layernorm_bwd')
    return dx, dxh


def geglu_fwd(h):
    """h (rows, 2F) half contiguous -> (rows, F) = h[:, :F] * gelu(h[:, F:])."""
    _dev(h)
    rows, F2 = h.shape
    assert h.is_contiguous() and F2 % 16 == 0
    y = torch.empty((rows, F2 // 2), dtype=h.dtype, device=h.device)
    L = _lib.load()
    _lib.check(L.mos_geglu_fwd(_p(h), _p(y), rows, F2 // 2, _dt(h), _stream()), 'mos_geglu_fwd')
    return y


def geglu_bwd(dy, h):
    _dev(dy, h)
    rows, F2 = h.shape
    assert h.is_contiguous() and dy.is_contiguous() and dy.shape == (rows, F2 // 2) and dy.dtype == h.dtype
    dh = torch.empty_like(h)
    L = _lib.load()
    _lib.check(L.mos_geglu_bwd(_p(dy), _p(h), _p(dh), rows, F2 // 2, _dt(h), _stream()), 'mos_geglu_bwd')
    return dh


def quick_gelu_fwd(x):
    """x * sigmoid(1.702 x) of a contiguous half tensor (numel % 8 == 0): the CLIP text tower's MLP activation."""
    _dev(x)
    assert x.is_contiguous() and x.numel() % 8 == 0
    y = torch.empty_like(x)
    L = _lib.load()
    _lib.check(L.mos_quick_gelu_fwd(_p(x), _p(y), x.numel(), _dt(x), _stream()), 'mos_quick_gelu_fwd')
    return y


def quick_gelu_bwd(dy, x):
    _dev(dy, x)
    assert x.is_contiguous() and dy.is_contiguous() and dy.shape == x.shape and dy.dtype == x.dtype
    dx = torch.empty_like(x)
    L = _lib.load()
    _lib.check(L.mos_quick_gelu_bwd(_p(dy), _p(x), _p(dx), x.numel(), _dt(x), _stream()), 'mos_quick_gelu_bwd')
    return dx


def softmax_rows(x, scale, out=None):
    """softmax(scale * x) over the last dim of a contiguous half (rows, N) tensor (in place when out is x)."""
    _dev(x)
    rows, N = x.shape
    assert x.is_contiguous()
    y = out if out is not None else torch.empty_like(x)
    L = _lib.load()
    _lib.check(L.mos_softmax_rows(_p(x), _p(y), rows, N, float(scale), _dt(x), _stream()), 'mos_softmax_rows')
    return y


def single_head_attention_nograd(q, k, v, scale):
    """softmax(scale q k^T) v for ONE head of large dim (VAE mid-block: d = 512, N = 4096), forward only: scores GEMM,
    row softmax, values GEMM per batch element on the library kernels (the (N, N) scores are materialised, 34 MB; above 8192
    queries -- the 1024 x 2048 images of the reference's regionally_sample.sh: N = 32768 -- in blocks of 8192 query rows)."""
    _dev(q, k, v)
    import math
    B, N, d = q.shape
    Nk = k.shape[1]
    assert Nk <= 32768 and Nk % 8 == 0, 'single_head_attention_nograd: up to 32768 keys (one softmax row per workgroup)'
    o = torch.empty_like(q)
    QB = 8192
    S = torch.empty((min(N, QB), Nk), dtype=q.dtype, device=q.device)
    # the scores are stored in half between the two GEMMs: q is pre-scaled by the power of two below `scale` (exact in
    # half) so that they are stored at (nearly) their softmax scale -- unscaled d = 512 scores are 22x larger, which
    # costs ~0.1 of a scaled logit in rounding at |logit| ~ 100 and overflows half from |logit| ~ 2900 on
    p2 = 2.0 ** math.floor(math.log2(scale)) if scale > 0 else 1.0
    qs = q * p2
    for i in range(B):
        vt = v[i].t().contiguous()
        for r0 in range(0, N, QB):
            r1 = min(N, r0 + QB)
            Sb = S[:r1 - r0]
            linear_fwd(qs[i, r0:r1], k[i], out=Sb)
            softmax_rows(Sb, scale / p2, out=Sb)
            linear_fwd(Sb, vt, out=o[i, r0:r1])
    return o


# ------------------------------------------------------------------------------------------------
# 3x3 convolution on channels-last activations (implicit GEMM) — caller-side operator
# ------------------------------------------------------------------------------------------------
def conv3x3_s2_nhwc(x, w_ohwi, bias=None, pad_mode=1, split_k=True):
    """3x3 / stride-2 convolution (forward): x (B, Cin, Hin, Win) half, channels_last; w_ohwi (Cout, 3, 3, Cin).
    pad_mode 1: padding 1 (UNet Downsample2D); 2: the VAE encoder's F.pad(x, (0, 1, 0, 1)) + padding 0, without the padded copy."""
    _dev(x, w_ohwi, bias)
    B, Cin, Hin, Win = x.shape
    assert _is_nhwc(x) or x.is_contiguous(memory_format=torch.channels_last), 'conv3x3_s2_nhwc needs channels_last'
    Cout = w_ohwi.shape[0]
    assert w_ohwi.shape == (Cout, 3, 3, Cin) and w_ohwi.is_contiguous() and w_ohwi.dtype == x.dtype and pad_mode in (1, 2)
    H, W = ((Hin - 1) // 2 + 1, (Win - 1) // 2 + 1) if pad_mode == 1 else ((Hin - 2) // 2 + 1, (Win - 2) // 2 + 1)
    y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    L = _lib.load()
    nbytes = L.mos_conv3x3_nhwc_workspace_bytes(B, H, W, Cin, Cout) if split_k else 0
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device) if nbytes > 0 else None
    _lib.check(L.mos_conv3x3_s2_nhwc(_p(x), _p(w_ohwi), _p(bias), _p(y), B, Hin, Win, Cin, Cout, int(pad_mode), _dt(x), _p(ws),
                                     _stream()), 'mos_conv3x3_s2_nhwc')
    return y


def conv3x3_nhwc(x, w_ohwi, bias=None, tbias=None, residual=None, upsample2x=False, split_k=True, gn_stats=False):
    """x: (B, Cin, H, W) half tensor in channels_last memory format, or a channel slice of one (`nhwc_pixel_stride`: read in place);
    w_ohwi: (Cout, 3, 3, Cin) contiguous half;
    bias fp32 (Cout,); tbias (B, Cout) half; residual like the output. Returns (B, Cout, H', W') channels_last
    (H' = 2H with upsample2x). split_k=False: no workspace is handed over, i.e. the unsplit kernel also on the
    low-resolution levels (tests). gn_stats=True: returns (y, chan_part) where chan_part is the (B, tiles, Cout, 2) fp32
    GroupNorm statistics of y from the kernel's epilogue, or None where this shape's kernel form keeps none."""
    _dev(x, w_ohwi, bias, tbias, residual)
    B, Cin, Hs, Ws = x.shape
    ps = nhwc_pixel_stride(x)
    assert ps is not None, 'conv3x3_nhwc needs channels_last (or a channel slice of a channels_last tensor)'
    Cout = w_ohwi.shape[0]
    assert w_ohwi.shape == (Cout, 3, 3, Cin) and w_ohwi.is_contiguous() and w_ohwi.dtype == x.dtype
    H, W = (2 * Hs, 2 * Ws) if upsample2x else (Hs, Ws)
    y = torch.empty((B, Cout, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.is_contiguous()
    if tbias is not None:
        assert tbias.shape == (B, Cout) and tbias.dtype == x.dtype and tbias.is_contiguous()
    if residual is not None:
        # (memory-format check, not a stride comparison: the stride of a size-1 dimension -- batch 1, a 1x1 map -- is arbitrary)
        assert residual.shape == y.shape and residual.dtype == x.dtype and residual.is_contiguous(memory_format=torch.channels_last)
    L = _lib.load()
    nbytes = L.mos_conv3x3_nhwc_workspace_bytes(B, H, W, Cin, Cout) if split_k else 0     # > 0: the split-K form of the low-resolution levels
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device) if nbytes > 0 else None
    part = None
    if gn_stats:
        tiles = L.mos_conv3x3_gn_tiles(B, H, W, Cin, Cout) if split_k else 0
        if tiles > 0:
            part = torch.empty((B, tiles, Cout, 2), dtype=torch.float32, device=x.device)
    _lib.check(L.mos_conv3x3_nhwc_px(_p(x), ps, _p(w_ohwi), _p(bias), _p(tbias), _p(residual), _p(y), B, H, W, Cin, Cout,
                                     int(bool(upsample2x)), _dt(x), _p(ws), _p(part), _stream()), 'mos_conv3x3_nhwc_px')
    return (y, part) if gn_stats else y
