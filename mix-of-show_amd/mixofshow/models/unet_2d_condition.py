"""SD-1.5 UNet2DConditionModel — the caller of every attention processor on the hot path.

diffusers is not available; this is a from-scratch implementation of the SD-1.5 configuration of
`UNet2DConditionModel` (block_out_channels (320, 640, 1280, 1280), 2 layers per block, 8 heads,
cross_attention_dim 768) whose module tree and state-dict keys match diffusers so real
SD-1.5 / ChilloutMix weights load unchanged and the reference's name-based logic keeps working
(`named_modules()` paths used as LoRA checkpoint keys, trainer_edlora.py:106-133; class names `Attention`,
`Transformer2DModel`; `down_blocks` / `mid_block` / `up_blocks` traversal order, edlora.py:186-189).

Non-attention operators (3x3 convs, GroupNorm, SiLU, GEGLU feed-forward) are plumbing and run on
PyTorch-ROCm (MIOpen / hipBLASLt); every `Attention` goes through its processor -> HIP kernels.
`cross_attention_kwargs` reach BOTH attn1 and attn2 of each block, and `down_block_additional_residuals`
(T2I-Adapter features) are consumed in the diffusers 0.19 order (pipeline_regionally_t2iadapter.py:556-566).
"""
import math
import os
import weakref
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from mixofshow.hip.functional import (add_layer_norm, conv1x1, conv3x3, conv3x3_stride2, geglu, group_norm_act, linear_geglu,  # noqa: F401
                                       linear_residual)
from mixofshow.models.attention import Attention


_freq_cache = {}


def _timestep_frequencies(half, downscale_freq_shift, max_period, device):
    """exp(-ln(max_period) * k / (half - shift)), k < half: a constant of the model, kept per device (4 tiny launches per UNet
    call otherwise). Not memoised while a hipGraph is being captured (the tensor would live in the graph's private pool)."""
    key = (half, float(downscale_freq_shift), float(max_period), str(device))
    f = _freq_cache.get(key)
    if f is None:
        exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=device)
        f = torch.exp(exponent / (half - downscale_freq_shift))
        if not (f.device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
            _freq_cache[key] = f
    return f


def get_timestep_embedding(timesteps, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, max_period=10000):
    half = dim // 2
    emb = timesteps[:, None].float() * _timestep_frequencies(half, downscale_freq_shift, max_period, timesteps.device)[None, :]
    # (diffusers: cat([sin, cos]) and, with flip_sin_to_cos, a second cat that swaps the halves -- the same values, one launch)
    return torch.cat([torch.cos(emb), torch.sin(emb)] if flip_sin_to_cos else [torch.sin(emb), torch.cos(emb)], dim=-1)


class TimestepEmbedding(nn.Module):

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


_act_memo = [None, None, None]


def _act_once(act, temb):
    """`act(temb)` for the time embedding that all 22 ResNet blocks of one UNet call share: computed by the first block, reused
    by the others (diffusers: one SiLU launch per block). UNet2DConditionModel.forward clears the memo when it returns, so
    the activation (and the autograd graph behind it) is not retained beyond its call."""
    ref, ver, out = _act_memo
    if ref is not None and ref() is temb and ver == temb._version and out.requires_grad == (
            temb.requires_grad and torch.is_grad_enabled()):
        return out
    out = act(temb)
    _act_memo[:] = [weakref.ref(temb), temb._version, out]
    return out


def _act_forget():
    _act_memo[:] = [None, None, None]


# Project the time embedding for ALL ResNet blocks of a UNet call with one `baddbmm` per output width instead of one tiny GEMM
# per block (M = batch: 22 launches of ~8.5 us in SD-1.5). Same-box A/B (profiles/r04_ab_same_box_*_switches.txt): training step
# 38.69 -> 38.41 ms, 512x768 regional sample 508.8 -> 501.9 ms. MOS_BATCH_TEMB=0: the per-block projections.
_batch_time_proj = os.environ.get('MOS_BATCH_TEMB', '1') != '0'


class _TimeProjections:
    """Stacked `time_emb_proj` weights of a model's ResNet blocks, grouped by output width: (blocks, C, temb) and (blocks, 1, C).
    The result of a group is (blocks, B, C), so each block's slice is a contiguous (B, C) tensor -- what the convolution
    epilogue consumes. Frozen projections only (the stacks are rebuilt when a weight's version or storage changes)."""

    def __init__(self, root):
        self.blocks = [m for m in root.modules() if isinstance(m, ResnetBlock2D) and m.time_emb_proj is not None]
        self.key, self.groups = None, []

    def usable(self):
        # (under no_grad -- every sampling pipeline -- ordinary requires_grad=True modules count as frozen, cf. functional._frozen)
        grad = torch.is_grad_enabled()
        return bool(self.blocks) and all(
            not (p.bias is None or (grad and (p.weight.requires_grad or p.bias.requires_grad)) or p._forward_hooks
                 or p._forward_pre_hooks) and type(p) is nn.Linear for p in (b.time_emb_proj for b in self.blocks))

    def __call__(self, act):
        key = tuple((b.time_emb_proj.weight.data_ptr(), b.time_emb_proj.weight._version, b.time_emb_proj.weight.dtype,
                     b.time_emb_proj.bias.data_ptr(), b.time_emb_proj.bias._version, b.time_emb_proj.bias.dtype) for b in self.blocks)
        if key != self.key:
            by_width = {}
            for b in self.blocks:
                by_width.setdefault(b.time_emb_proj.out_features, []).append(b)
            self.groups = [(bs, torch.stack([b.time_emb_proj.weight.detach() for b in bs]).transpose(1, 2).contiguous(),
                            torch.stack([b.time_emb_proj.bias.detach() for b in bs]).unsqueeze(1)) for bs in by_width.values()]
            self.key = key
        out = {}
        for bs, w, bias in self.groups:       # (dtypes as for nn.Linear: equal, or reconciled by autocast)
            y = torch.baddbmm(bias, act.unsqueeze(0).expand(len(bs), *act.shape), w)
            for i, b in enumerate(bs):
                out[id(b)] = y[i]
        return out


class ResnetBlock2D(nn.Module):

    def __init__(self, in_channels, out_channels, temb_channels=1280, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb=None):
        # on the HIP device: fused GroupNorm+SiLU kernels, implicit-GEMM 3x3 convolutions with the time-embedding add and
        # the residual add in their epilogues (diffusers: conv, + temb[:, :, None, None], ..., x + h as separate kernels)
        tb = None
        if self.time_emb_proj is not None and temb is not None:
            pre = getattr(temb, '_mos_time_bias', None)          # UNet2DConditionModel.forward projected all blocks at once
            tb = pre.get(id(self)) if pre is not None else None
            if tb is None:
                tb = self.time_emb_proj(_act_once(self.nonlinearity, temb))
        # (x feeds norm1 AND the skip path: taking the skip from the norm's tap adds its gradient inside the norm's backward)
        x, h = group_norm_act(self.norm1, x, True, tap=True)
        # (gn_groups: conv1's output feeds norm2, conv2's the GroupNorm of whatever follows the block -- the next resnet's norm1
        #  or a Transformer2DModel's norm, all of this module family's group count; on large maps the convolution's epilogue
        #  leaves that norm's statistics with its output, mixofshow.hip.functional._attach_gn_stats)
        h = conv3x3(self.conv1, h, tbias=tb, gn_groups=self.norm2.num_groups)
        if self.conv_shortcut is not None:
            x = conv1x1(self.conv_shortcut, x)
        return conv3x3(self.conv2, self.dropout(group_norm_act(self.norm2, h, True)), residual=x, gn_groups=self.norm1.num_groups)


class Downsample2D(nn.Module):

    def __init__(self, channels, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        # VAE encoder (padding 0): the asymmetric F.pad(x, (0, 1, 0, 1)) is folded into the kernel's bounds on the HIP path
        return conv3x3_stride2(self.conv, x, pad_bottom_right=self.padding == 0)


class Upsample2D(nn.Module):

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return conv3x3(self.conv, x, upsample=True)      # the nearest 2x upsample is folded into the conv's input addressing


class GEGLU(nn.Module):

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        # value * gelu(gate). Sampling: formed in the epilogue of the projection GEMM (the (rows, 8C) pre-activation never
        # reaches HBM); training: GEMM + one fused kernel each way (the backward needs the pre-activation)
        return linear_geglu(self.proj, x)


class FeedForward(nn.Module):

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])

    def forward(self, x, residual=None):
        """`net(x)` (+ residual: the transformer block's `ff(norm3(h)) + h`, added in the epilogue of the last GEMM)."""
        x = self.net[1](self.net[0](x))
        if residual is None:
            return self.net[2](x)
        return linear_residual(self.net[2], x, residual)


_attn_out_residual = os.environ.get('MOS_ATTN_OUT_RESIDUAL', '1') != '0'      # host-side A/B switch (read once)


def _attn_plus_stream(attn, n, x, encoder_hidden_states, cak):
    """(attn(n) + x, True) with the add inside the out-projection GEMM's epilogue when the layer's processor runs
    mixofshow.models.attention.fused_attention_layer on half CUDA tensors; (attn(n), False) otherwise -- any other processor
    never looks at the side channel, and the caller adds in the LayerNorm kernel as before (same rounding points either way)."""
    offer = _attn_out_residual and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.shape == n.shape
    if offer:
        attn.__dict__['_mos_residual'] = x
        attn.__dict__.pop('_mos_residual_fused', None)
    a = attn(n, encoder_hidden_states=encoder_hidden_states, **cak)
    if not offer:
        return a, False
    attn.__dict__.pop('_mos_residual', None)
    return a, bool(attn.__dict__.pop('_mos_residual_fused', False))


class BasicTransformerBlock(nn.Module):

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        cak = cross_attention_kwargs if cross_attention_kwargs is not None else {}
        # x = attn1(norm1(x)) + x; x = attn2(norm2(x)) + x; x = ff(norm3(x)) + x  with each residual sum formed inside the
        # LayerNorm kernel that consumes it, and each bypass gradient added inside that norm's backward kernel
        # Round 6: where the attention layer runs on the fused HIP path its out-projection GEMM adds the residual stream in its
        # epilogue (`_attn_plus_stream`), so the next LayerNorm takes the sum as its only input.
        x, n = add_layer_norm(self.norm1, x)
        a, summed = _attn_plus_stream(self.attn1, n, x, None, cak)
        x, n = add_layer_norm(self.norm2, a) if summed else add_layer_norm(self.norm2, x, a)
        a, summed = _attn_plus_stream(self.attn2, n, x, encoder_hidden_states, cak)
        x, n = add_layer_norm(self.norm3, a) if summed else add_layer_norm(self.norm3, x, a)
        return self.ff(n, residual=x)


class Transformer2DModel(nn.Module):

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states=None, cross_attention_kwargs=None):
        b, c, h, w = x.shape
        residual, x = group_norm_act(self.norm, x, False, tap=True)
        x = conv1x1(self.proj_in, x)
        x = x.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        for blk in self.transformer_blocks:
            x = blk(x, encoder_hidden_states=encoder_hidden_states, cross_attention_kwargs=cross_attention_kwargs)
        # (b, hw, c) -> NCHW view with channels-last strides: free when the UNet runs in channels_last memory format
        # (the token-major layout of the attention path IS NHWC); the 1x1 conv accepts either layout
        x = x.reshape(b, h, w, -1).permute(0, 3, 1, 2)
        return conv1x1(self.proj_out, x, residual=residual)


class CrossAttnDownBlock2D(nn.Module):
    has_cross_attention = True

    def __init__(self, in_channels, out_channels, temb, heads, cross_dim, add_downsample, layers=2):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb) for i in range(layers)])
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, out_channels // heads, out_channels, cross_dim) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, x, temb, encoder_hidden_states=None, cross_attention_kwargs=None, additional_residuals=None):
        outs = ()
        n = len(self.resnets)
        for i, (res, att) in enumerate(zip(self.resnets, self.attentions)):
            x = att(res(x, temb), encoder_hidden_states, cross_attention_kwargs)
            if i == n - 1 and additional_residuals is not None:
                x = x + additional_residuals
            outs += (x, )
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x, )
        return x, outs


class DownBlock2D(nn.Module):
    has_cross_attention = False

    def __init__(self, in_channels, out_channels, temb, add_downsample, layers=2):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, temb) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, x, temb):
        outs = ()
        for res in self.resnets:
            x = res(x, temb)
            outs += (x, )
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x, )
        return x, outs


class UNetMidBlock2DCrossAttn(nn.Module):

    def __init__(self, channels, temb, heads, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(channels, channels, temb), ResnetBlock2D(channels, channels, temb)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, channels // heads, channels, cross_dim)])

    def forward(self, x, temb, encoder_hidden_states=None, cross_attention_kwargs=None):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, encoder_hidden_states, cross_attention_kwargs)
        return self.resnets[1](x, temb)


class _UpBase(nn.Module):

    def _resnets(self, in_channels, prev_channels, out_channels, temb, layers):
        blocks = []
        for i in range(layers):
            skip = in_channels if i == layers - 1 else out_channels
            rin = prev_channels if i == 0 else out_channels
            blocks.append(ResnetBlock2D(rin + skip, out_channels, temb))
        return nn.ModuleList(blocks)


class UpBlock2D(_UpBase):
    has_cross_attention = False

    def __init__(self, in_channels, prev_channels, out_channels, temb, add_upsample, layers=3):
        super().__init__()
        self.resnets = self._resnets(in_channels, prev_channels, out_channels, temb, layers)
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, x, res_tuple, temb):
        for res in self.resnets:
            x = res(torch.cat([x, res_tuple[-1]], dim=1), temb)
            res_tuple = res_tuple[:-1]
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class CrossAttnUpBlock2D(_UpBase):
    has_cross_attention = True

    def __init__(self, in_channels, prev_channels, out_channels, temb, heads, cross_dim, add_upsample, layers=3):
        super().__init__()
        self.resnets = self._resnets(in_channels, prev_channels, out_channels, temb, layers)
        self.attentions = nn.ModuleList(
            [Transformer2DModel(heads, out_channels // heads, out_channels, cross_dim) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, x, res_tuple, temb, encoder_hidden_states=None, cross_attention_kwargs=None):
        for res, att in zip(self.resnets, self.attentions):
            x = res(torch.cat([x, res_tuple[-1]], dim=1), temb)
            res_tuple = res_tuple[:-1]
            x = att(x, encoder_hidden_states, cross_attention_kwargs)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetOutput(SimpleNamespace):
    pass


SD15_UNET_CONFIG = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                        layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32)


class UNet2DConditionModel(nn.Module):

    def __init__(self, sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32):
        super().__init__()
        self.config = SimpleNamespace(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      attention_head_dim=attention_head_dim, cross_attention_dim=cross_attention_dim,
                                      norm_num_groups=norm_num_groups)
        self.in_channels = in_channels
        heads = attention_head_dim  # SD-1.5 quirk: `attention_head_dim` is the NUMBER of heads
        ch = list(block_out_channels)
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self._time_dim = ch[0]
        n = len(ch)
        downs = []
        out_c = ch[0]
        for i in range(n):
            in_c, out_c = out_c, ch[i]
            last = i == n - 1
            if not last:
                downs.append(CrossAttnDownBlock2D(in_c, out_c, temb, heads, cross_attention_dim, True, layers_per_block))
            else:
                downs.append(DownBlock2D(in_c, out_c, temb, False, layers_per_block))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = UNetMidBlock2DCrossAttn(ch[-1], temb, heads, cross_attention_dim)
        rev = ch[::-1]
        ups = []
        out_c = rev[0]
        for i in range(n):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, n - 1)]
            last = i == n - 1
            if i == 0:
                ups.append(UpBlock2D(in_c, prev_c, out_c, temb, not last, layers_per_block + 1))
            else:
                ups.append(CrossAttnUpBlock2D(in_c, prev_c, out_c, temb, heads, cross_attention_dim, not last,
                                              layers_per_block + 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, ch[0], eps=1e-5)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)
        self.gradient_checkpointing = False
        self.channels_last = False

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        # `.to(memory_format=torch.channels_last)` converts the conv weights; remember it so that forward() puts the input
        # latents into the same layout: NHWC end to end = no NCHW<->NHWC transposes around MIOpen's convolutions and no
        # permute copies around the transformer blocks (token-major IS channels-last)
        w = self.conv_in.weight
        self.channels_last = bool(w.dim() == 4 and w.shape[1] > 1 and not w.is_contiguous()
                                  and w.is_contiguous(memory_format=torch.channels_last))
        return out

    def enable_gradient_checkpointing(self):
        self.gradient_checkpointing = True

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                down_block_additional_residuals=None, return_dict=True):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=sample.device)
        elif timestep.dim() == 0:
            timestep = timestep[None].to(sample.device)
        timestep = timestep.expand(sample.shape[0])
        t_emb = get_timestep_embedding(timestep, self._time_dim).to(dtype=sample.dtype)
        emb = self.time_embedding(t_emb)
        if _batch_time_proj and (sample.is_cuda or _batch_time_proj == 'force'):    # a launch-count optimisation of the device path
            tp = self.__dict__.get('_time_projections')
            if tp is None:
                tp = _TimeProjections(self)
                object.__setattr__(self, '_time_projections', tp)
            if tp.usable():
                emb._mos_time_bias = tp(_act_once(self.down_blocks[0].resnets[0].nonlinearity, emb))

        if torch.is_tensor(encoder_hidden_states) and encoder_hidden_states.dim() == 4 and sample.is_cuda:
            from mixofshow.models import edlora
            attn2 = self.__dict__.get('_attn2_modules')
            if attn2 is None:
                attn2 = [m for n, m in self.named_modules() if isinstance(m, Attention) and n.endswith('attn2')]
                object.__setattr__(self, '_attn2_modules', attn2)
            if any(isinstance(m.processor, (edlora.EDLoRA_AttnProcessor, edlora.EDLoRA_Control_AttnProcessor)) for m in attn2):
                edlora.attach_layer_major_states(encoder_hidden_states)   # one transpose instead of 16 strided gathers

        if getattr(self, 'channels_last', False):
            sample = sample.contiguous(memory_format=torch.channels_last)
        sample = self.conv_in(sample)
        res = (sample, )
        is_adapter = down_block_additional_residuals is not None
        for blk in self.down_blocks:
            if blk.has_cross_attention:
                extra = None
                if is_adapter and len(down_block_additional_residuals) > 0:
                    extra = down_block_additional_residuals.pop(0)
                sample, outs = self._run(blk, sample, emb, encoder_hidden_states, cross_attention_kwargs, extra)
            else:
                sample, outs = blk(sample, emb)
                if is_adapter and len(down_block_additional_residuals) > 0:
                    sample += down_block_additional_residuals.pop(0)  # in place, as diffusers 0.19 (also hits the skip)
            res += outs
        sample = self.mid_block(sample, emb, encoder_hidden_states, cross_attention_kwargs)
        for blk in self.up_blocks:
            k = len(blk.resnets)
            take, res = res[-k:], res[:-k]
            if blk.has_cross_attention:
                sample = blk(sample, take, emb, encoder_hidden_states, cross_attention_kwargs)
            else:
                sample = blk(sample, take, emb)
        sample = self.conv_out(group_norm_act(self.conv_norm_out, sample, True))
        _act_forget()
        return UNetOutput(sample=sample) if return_dict else (sample, )

    def _run(self, blk, sample, emb, ehs, cak, extra):
        if self.gradient_checkpointing and self.training and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            return checkpoint(blk, sample, emb, ehs, cak, extra, use_reentrant=False)
        return blk(sample, emb, ehs, cak, extra)
