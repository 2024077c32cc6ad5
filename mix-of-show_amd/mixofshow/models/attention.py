"""`Attention` — the module whose attention-processor protocol is the plug-in boundary of the hot path.

diffusers is not a dependency of this framework; this class restates the slice of
`diffusers.models.attention_processor.Attention` (0.19.x) that the reference's processors touch
(reference mixofshow/models/edlora.py:40-98; pipeline_regionally_t2iadapter.py:88-145):
attributes `to_q/to_k/to_v/to_out/heads/scale/upcast_*/spatial_norm/group_norm/norm_cross/
residual_connection/rescale_output_factor` and the helper methods `head_to_batch_dim`,
`batch_to_head_dim`, `get_attention_scores`, `prepare_attention_mask`, `set_processor`.
State-dict keys match diffusers (`to_q.weight`, `to_out.0.bias`, ...), so real SD-1.5 weights load.

The helper methods are plain torch and exist so that third-party processors written against the
diffusers protocol run unchanged; the framework's own processors (MosAttnProcessor here, the EDLoRA /
regional processors in mixofshow.models.edlora and mixofshow.pipelines) never call them — they go
through the fused HIP path in `fused_attention_layer`.
"""
import torch
import torch.nn as nn

from mixofshow.hip import functional as F_hip


class Attention(nn.Module):

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, out_bias=True, processor=None):
        super().__init__()
        inner_dim = dim_head * heads
        self.query_dim = query_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.heads = heads
        self.scale = dim_head**-0.5
        self.rescale_output_factor = 1.0
        self.residual_connection = False
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.to_q = nn.Linear(query_dim, inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim, bias=out_bias), nn.Dropout(dropout)])
        object.__setattr__(self, '_mos_cache', F_hip.WeightCache())
        self.processor = None
        self.set_processor(processor if processor is not None else MosAttnProcessor())

    # ---- protocol -------------------------------------------------------------------------
    def set_processor(self, processor):
        if isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop('processor')
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **cross_attention_kwargs)

    def head_to_batch_dim(self, tensor):
        b, s, c = tensor.shape
        h = self.heads
        return tensor.reshape(b, s, h, c // h).permute(0, 2, 1, 3).reshape(b * h, s, c // h)

    def batch_to_head_dim(self, tensor):
        bh, s, d = tensor.shape
        h = self.heads
        return tensor.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            scores = torch.bmm(query, key.transpose(-1, -2)) * self.scale
        else:
            scores = torch.baddbmm(attention_mask, query, key.transpose(-1, -2), beta=1, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None):
        if attention_mask is None:
            return None
        raise NotImplementedError('attention masks are not used on the SD-1.5 path (reference passes None)')


def _lora_site(linear):
    lora = getattr(linear, '_mos_lora', None)
    if lora is None:
        return None
    return (lora.lora_down.weight, lora.lora_up.weight, lora.alpha_value())


def _sites(*linears):
    s = [_lora_site(l) for l in linears]
    if all(x is None for x in s):
        return []
    if any(x is None for x in s):
        return None  # mixed: caller falls back to per-projection calls
    if len(s) > 1 and sum(x[0].shape[0] for x in s) > F_hip.MAX_PACKED_RANK:
        return None  # ranks do not fit one packed rank-16 operand together: one GEMM per projection instead
    return s


def project(attn, name, linears, x, cd, residual=None):
    """Fused projection of x through 1..3 Linear layers of `attn` (+ their LoRA branches): one GEMM (+ `residual` added to the
    rounded result in its epilogue: bit-identical to the GEMM followed by a half add)."""
    need_wt = torch.is_grad_enabled() and (x.requires_grad or any(
        getattr(l, '_mos_lora', None) is not None for l in linears))
    W16, Wt16 = attn._mos_cache.weight(name, [l.weight for l in linears], cd, transposed=need_wt)
    b32 = attn._mos_cache.bias(name, [l.bias for l in linears])
    sites = _sites(*linears)
    assert sites is not None
    y = F_hip.lora_linear(x, W16, Wt16, b32, sites, residual=residual)
    tap = getattr(attn, '_mos_tap', None)
    assert tap is None or residual is None
    if tap is not None:
        # feature tap (gradient fusion): the fused GEMM bypasses nn.Linear.__call__, so forward hooks on
        # to_q/to_k/to_v/to_out.0 would never fire — report (module, input, output) per projection instead
        off = 0
        for l in linears:
            n = l.weight.shape[0]
            tap(l, x, y[..., off:off + n])
            off += n
    return y


def fused_attention_layer(attn, hidden_states, encoder_hidden_states=None, tok_idx=None, region=None, edit_probs=None):
    """The whole attention layer on the HIP path: projections (+LoRA), fused attention, out-projection.

    hidden_states (B, N, C); encoder_hidden_states (B, M, Cc) or None (self-attention).
    tok_idx: int32 (B, T) key positions whose probabilities are returned (training regulariser).
    region: None or dict(k_src, v_src, boxes, feat_h, feat_w) -> regional mask-and-blend attention.
    edit_probs: None, or a callable P -> P' for the materialised-probability split of a CROSS layer (the reference's
    `attention_probs = self.controller(attention_probs, ...)` between softmax and bmm, edlora.py:81-83): P is the dense
    (B*H, N, M) tensor of mos_attn_probs, P' (the same object, edited in place, or a new tensor) feeds mos_attn_pv.
    Returns (out (B, N, C), pcols (B, H, N, T) fp32 | None)."""
    # (device guard: every primitive in mixofshow.hip.ops raises on non-HIP tensors — there is no fallback)
    cd = F_hip.compute_dtype_for(hidden_states)
    out_dtype = cd if (torch.is_autocast_enabled('cuda') or hidden_states.dtype in (torch.float16, torch.bfloat16)) \
        else hidden_states.dtype
    x = hidden_states if hidden_states.dtype == cd else hidden_states.to(cd)
    pcols = None
    # BasicTransformerBlock's `attn(norm(h)) + h` (round 6): the block leaves h here and the out-projection GEMM adds it in its
    # epilogue, so the LayerNorm that follows reads ONE tensor and writes one. Taken only where it is exact: same dtype and
    # shape as the layer output, no feature tap on the projections (gradient fusion records the bare to_out.0 output), no
    # output rescaling. A processor that never gets here leaves the attribute in place and the block adds as before.
    residual = attn.__dict__.pop('_mos_residual', None)
    if residual is not None and not (residual.dtype == cd and out_dtype == cd and residual.shape == hidden_states.shape
                                     and getattr(attn, '_mos_tap', None) is None and attn.rescale_output_factor == 1.0
                                     and not attn.residual_connection):
        residual = None
    if encoder_hidden_states is None:
        if _sites(attn.to_q, attn.to_k, attn.to_v) is not None:
            qkv = project(attn, 'qkv', [attn.to_q, attn.to_k, attn.to_v], x, cd)
            o = F_hip.attention_qkv(qkv, attn.heads, attn.scale)
        else:
            q = project(attn, 'q', [attn.to_q], x, cd)
            k = project(attn, 'k', [attn.to_k], x, cd)
            v = project(attn, 'v', [attn.to_v], x, cd)
            o, _ = F_hip.attention(q, k, v, attn.heads, attn.scale)
    else:
        e = encoder_hidden_states if encoder_hidden_states.dtype == cd else encoder_hidden_states.to(cd)
        q = project(attn, 'q', [attn.to_q], x, cd)
        if region is not None:
            from mixofshow.hip import ops
            o = ops.region_attn_fwd(q, region['k_src'], region['v_src'], attn.heads, attn.scale, region['boxes'],
                                    region['feat_h'], region['feat_w'])
        elif edit_probs is not None:
            o = _attention_through_probs(attn, q, e, cd, edit_probs)
        elif _sites(attn.to_k, attn.to_v) is not None:
            kv = project(attn, 'kv', [attn.to_k, attn.to_v], e, cd)
            o, pcols = F_hip.attention_q_kv(q, kv, attn.heads, attn.scale, tok_idx=tok_idx)
        else:
            k = project(attn, 'k', [attn.to_k], e, cd)
            v = project(attn, 'v', [attn.to_v], e, cd)
            o, pcols = F_hip.attention(q, k, v, attn.heads, attn.scale, tok_idx=tok_idx)
    out = project(attn, 'out', [attn.to_out[0]], o, cd, residual=residual)
    if residual is not None:
        attn.__dict__['_mos_residual_fused'] = True
    out = attn.to_out[1](out)
    if out.dtype != out_dtype:
        out = out.to(out_dtype)
    return out, pcols


def _attention_through_probs(attn, q, e, cd, edit_probs):
    """softmax(scale q k^T) as a tensor -> edit_probs -> P' v (mos_attn_probs / mos_attn_pv): the controller boundary of the
    reference with the full map, for controllers that do not declare the columns they read. Differentiable since round 6
    (mos_attn_probs_bwd / mos_attn_pv_bwd): the map reaches the controller WITH grad, as in the reference (edlora.py:81-83),
    so a reference-side `AttentionStore(training=True)` object trains."""
    if e.shape[1] > 96:
        raise NotImplementedError(f'materialised probabilities are built for text keys (<= 96), got {e.shape[1]}')
    if _sites(attn.to_k, attn.to_v) is not None:
        kv = project(attn, 'kv', [attn.to_k, attn.to_v], e, cd)
        C = kv.shape[-1] // 2
        k, v = kv[..., :C], kv[..., C:]
    else:
        k = project(attn, 'k', [attn.to_k], e, cd)
        v = project(attn, 'v', [attn.to_v], e, cd)
    probs = F_hip.attn_probs(q, k, attn.heads, attn.scale)
    edited = edit_probs(probs)
    probs = probs if edited is None else edited
    if probs.dtype != v.dtype:
        probs = probs.to(v.dtype)
    return F_hip.attn_pv(probs, v, attn.heads)


def _check_plain(attn):
    assert attn.spatial_norm is None and attn.group_norm is None and not attn.norm_cross, \
        'spatial_norm / group_norm / norm_cross are not part of the SD-1.5 transformer blocks'
    # upcast_attention / upcast_softmax (honoured by the reference, pipeline_regionally_t2iadapter.py:63-73, and by diffusers'
    # get_attention_scores) ask for fp32 scores / an fp32 softmax with P cast back to the layer dtype. That IS the arithmetic of
    # the fused kernels (scores and softmax statistics in fp32 registers, P rounded once to the half MFMA operand), so both flags
    # are accepted and change nothing; with the flags off the kernels are MORE exact than the reference's half scores, not less.


class MosAttnProcessor:
    """Default processor: fused HIP attention (what diffusers' AttnProcessor2_0 is to the reference's attn1)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kwargs):
        _check_plain(attn)
        assert attention_mask is None
        residual = hidden_states
        ndim = hidden_states.ndim
        if ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        out, _ = fused_attention_layer(attn, hidden_states, encoder_hidden_states)
        if ndim == 4:
            out = out.transpose(-1, -2).reshape(b, c, h, w)
        if attn.residual_connection:
            out = out + residual
        if attn.rescale_output_factor != 1.0:
            out = out / attn.rescale_output_factor
        return out
