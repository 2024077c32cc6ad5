"""ED-LoRA model surgery on the HIP path — same names and contracts as the reference's
mixofshow/models/edlora.py (LoRALinearLayer :221-246, EDLoRA_AttnProcessor :103-173,
EDLoRA_Control_AttnProcessor :22-100, revise_* installers :176-218, remove_* :12-19).

Differences that matter for speed, not for results:
  * LoRALinearLayer does not launch 3 GEMMs + scale + add; it registers itself on the wrapped module
    (`_mos_lora`) so that the attention processors fold every LoRA branch of a layer into the fused
    projection GEMMs (mixofshow.models.attention.fused_attention_layer). Called on its own (CLIP
    q/k/v/out_proj, ff / proj_in / proj_out sites) it runs the same fused kernel for a single site.
  * the control processor does not hand a (B*H, N, 77) probability tensor to the controller: it asks
    the controller which key positions it needs (`controller.token_positions`) and passes the
    (B, H, N, T) probabilities of those columns, computed inside the attention kernel.
"""
import math

import torch
import torch.nn as nn

from mixofshow.hip import functional as F_hip
from mixofshow.models.attention import MosAttnProcessor, _check_plain, fused_attention_layer


def remove_edlora_unet_attention_forward(unet):
    """reference edlora.py:12-19 — restore the default processor on every attn2."""

    def visit(module):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and name == 'attn2':
                layer.set_processor(MosAttnProcessor())
            else:
                visit(layer)

    visit(unet)


def attach_layer_major_states(states):
    """ED-LoRA's (B, L, 77, C) layer-wise text states get a LAYER-major copy attached (`states._mos_layers`, one contiguous
    (B, 77, C) tensor per cross-attention layer) for the processors of this module; the tensor itself is passed on unchanged,
    so processors written against the diffusers protocol keep indexing it.

    Indexing the batch-major tensor per layer (`states[:, k]`, reference :56-59) costs, per layer, a strided gather in front of
    the K/V projection GEMM and -- in training -- a zero-filled (B, L, 77, C) gradient buffer, a strided scatter and an
    accumulation (3 launches x 16 layers + 16 gathers per step). One transpose copy makes every layer's slice contiguous and
    `unbind` gives autograd ONE stack for the gradient of all slices. Cached on the tensor (sampling loops pass the same
    prompt embedding every step) and keyed by its version counter."""
    ent = getattr(states, '_mos_layers', None)
    if ent is None or ent[0] != states._version or ent[2] != (states.requires_grad and torch.is_grad_enabled()):
        ent = (states._version, states.transpose(0, 1).contiguous().unbind(0), states.requires_grad and torch.is_grad_enabled())
        states._mos_layers = ent
    return states


def refresh_layer_major_states(states):
    """After an IN-PLACE update of `states` (a sampling pipeline refilling the static prompt embedding a captured hipGraph
    reads): rewrite the attached layer-major copy in place as well -- the graph holds the addresses of its slices -- and
    re-key it to the new version. No-op when nothing is attached."""
    ent = getattr(states, '_mos_layers', None)
    if ent is None:
        return states
    with torch.no_grad():
        for dst, src in zip(ent[1], states.transpose(0, 1).unbind(0)):
            dst.copy_(src)
    states._mos_layers = (states._version, ent[1], ent[2])
    return states


def _select_layer_states(encoder_hidden_states, idx):
    # (B, 16, 77, 768) layer-wise ED-LoRA embedding -> this layer's slice (reference :56-59, :130-133)
    if encoder_hidden_states is not None and encoder_hidden_states.dim() == 4:
        ent = getattr(encoder_hidden_states, '_mos_layers', None)
        if ent is not None and ent[0] == encoder_hidden_states._version:
            return ent[1][idx]
        return encoder_hidden_states[:, idx]
    return encoder_hidden_states


def _run(attn, hidden_states, encoder_hidden_states, tok_idx=None, edit_probs=None):
    _check_plain(attn)
    residual = hidden_states
    ndim = hidden_states.ndim
    if ndim == 4:
        b, c, h, w = hidden_states.shape
        hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
    out, pcols = fused_attention_layer(attn, hidden_states, encoder_hidden_states, tok_idx=tok_idx, edit_probs=edit_probs)
    if ndim == 4:
        out = out.transpose(-1, -2).reshape(b, c, h, w)
    if attn.residual_connection:
        out = out + residual
    if attn.rescale_output_factor != 1.0:
        out = out / attn.rescale_output_factor
    return out, pcols


class EDLoRA_AttnProcessor:
    """Cross/self attention with the layer-indexed text states (reference edlora.py:103-173)."""

    def __init__(self, cross_attention_idx, attention_op=None):
        self.attention_op = attention_op
        self.cross_attention_idx = cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        assert attention_mask is None, 'the SD-1.5 path never passes an attention mask'
        ehs = _select_layer_states(encoder_hidden_states, self.cross_attention_idx)
        return _run(attn, hidden_states, ehs)[0]


class EDLoRA_Control_AttnProcessor:
    """Same, reporting attention probabilities to a controller (reference edlora.py:22-100).

    Three kinds of controller (the reference knows one: it always materialises the (B*H, N, 77) map, :81-83):
      * controllers that expose `token_positions` (int32 (B, T) key indices; mixofshow.utils.ptp_util.AttentionStore, the
        training-time regulariser) receive the (B, H, N, T) probabilities of exactly those columns, produced inside the fused
        kernel, with autograd through them;
      * `is_passthrough = True` (the reference's DummyController / EmptyControl): called with None, bookkeeping only;
      * ANY OTHER controller -- e.g. the reference's own `AttentionStore` / `AttentionControl` objects
        (mixofshow/utils/ptp_util.py:22-108), prompt-to-prompt editors -- gets the reference's protocol verbatim: the dense
        (B*H, N, 77) probability tensor (mos_attn_probs), may store it or edit it in place (the eval-mode rule "second half of
        the CFG batch only", ptp_util.py:45-46, is the controller's own code), and what it returns goes into P.V (mos_attn_pv).
        With autograd since round 6 (mos_attn_probs_bwd / mos_attn_pv_bwd): the map carries grad like the reference's, so its
        own AttentionStore(training=True) + a loss on the stored maps trains; declaring `token_positions` remains the fast
        path (no dense map)."""

    def __init__(self, cross_attention_idx, place_in_unet, controller, attention_op=None):
        self.cross_attention_idx = cross_attention_idx
        self.place_in_unet = place_in_unet
        self.controller = controller
        self.attention_op = attention_op

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        assert attention_mask is None
        is_cross = encoder_hidden_states is not None
        ehs = _select_layer_states(encoder_hidden_states, self.cross_attention_idx)
        ctrl = self.controller
        tok = getattr(ctrl, 'token_positions', None) if is_cross else None
        passthrough = getattr(ctrl, 'is_passthrough', False)
        if is_cross and tok is None and not passthrough:
            place = self.place_in_unet
            return _run(attn, hidden_states, ehs, edit_probs=lambda p: ctrl(p, True, place))[0]
        out, pcols = _run(attn, hidden_states, ehs, tok_idx=tok)
        if pcols is not None:
            ctrl(pcols, is_cross, self.place_in_unet)
        elif passthrough:
            ctrl(None, is_cross, self.place_in_unet)
        return out


def revise_edlora_unet_attention_forward(unet):
    """Install EDLoRA_AttnProcessor on every attn2, numbering layers 0..15 down->mid->up
    (reference edlora.py:176-190; the order is part of the checkpoint contract: embedding row k <-> layer k)."""

    def visit(module, count):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and 'attn2' in name:
                layer.set_processor(EDLoRA_AttnProcessor(count))
                count += 1
            else:
                count = visit(layer, count)
        return count

    idx = visit(unet.down_blocks, 0)
    idx = visit(unet.mid_block, idx)
    idx = visit(unet.up_blocks, idx)
    return idx


class _PassthroughController:
    is_passthrough = True

    def __init__(self):
        self.num_att_layers = 0

    def __call__(self, *args):
        return args[0]


def revise_edlora_unet_attention_controller_forward(unet, controller):
    """reference edlora.py:193-218 (controller None -> pass-through, 'down'/'mid'/'up' places)."""
    if controller is None:
        controller = _PassthroughController()

    def visit(module, count, place):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and 'attn2' in name:
                layer.set_processor(EDLoRA_Control_AttnProcessor(count, place, controller))
                count += 1
            else:
                count = visit(layer, count, place)
        return count

    idx = visit(unet.down_blocks, 0, 'down')
    idx = visit(unet.mid_block, idx, 'mid')
    idx = visit(unet.up_blocks, idx, 'up')
    controller.num_att_layers = idx
    return idx


class LoRALinearLayer(nn.Module):
    """Low-rank side branch on a Linear or 1x1 Conv2d (reference edlora.py:221-246).

    Same constructor, parameter names (`lora_down.weight`, `lora_up.weight`), init (kaiming-uniform
    a=sqrt(5) / zeros) and `alpha` buffer as the reference; replaces `original_module.forward` in place."""

    def __init__(self, name, original_module, rank=4, alpha=1):
        super().__init__()
        self.name = name
        self.is_conv = original_module.__class__.__name__ == 'Conv2d'
        if self.is_conv:
            cin, cout = original_module.in_channels, original_module.out_channels
            assert original_module.kernel_size == (1, 1), 'only 1x1 convolutions carry LoRA (trainer_edlora.py:130)'
            self.lora_down = nn.Conv2d(cin, rank, (1, 1), bias=False)
            self.lora_up = nn.Conv2d(rank, cout, (1, 1), bias=False)
        else:
            cin, cout = original_module.in_features, original_module.out_features
            self.lora_down = nn.Linear(cin, rank, bias=False)
            self.lora_up = nn.Linear(rank, cout, bias=False)
        self.register_buffer('alpha', torch.tensor(alpha))
        # host copy of the buffer (reading it per call would be a device sync); refreshed whenever the buffer object
        # or its version changes (load_state_dict / .to() / in-place writes), see alpha_value()
        self._alpha_value = float(alpha)
        self._alpha_tag = None
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        # keep a handle on the wrapped module without registering it as a child (no state-dict keys, no cycle
        # through nn.Module bookkeeping); the fused attention processors look the branch up through `_mos_lora`
        object.__setattr__(self, '_wrapped', original_module)
        object.__setattr__(self, '_cache', F_hip.WeightCache())
        self.original_forward = original_module.forward
        object.__setattr__(original_module, '_mos_lora', self)
        original_module.forward = self.forward

    def alpha_value(self):
        """Host value of the `alpha` buffer; one device read after each change of the buffer, none in steady state."""
        a = self.alpha
        tag = (a.data_ptr(), a._version)
        if tag != self._alpha_tag:
            if self._alpha_tag is not None or a.device.type != 'cpu':
                self._alpha_value = float(a.item())
            self._alpha_tag = tag
        return self._alpha_value

    def forward(self, hidden_states):
        mod = self._wrapped
        cd = F_hip.compute_dtype_for(hidden_states)
        out_dtype = cd if (torch.is_autocast_enabled('cuda') or hidden_states.dtype == cd) else hidden_states.dtype
        need_wt = torch.is_grad_enabled()
        W16, Wt16 = self._cache.weight('w', [mod.weight], cd, transposed=need_wt)
        b32 = self._cache.bias('w', [mod.bias])
        site = [(self.lora_down.weight, self.lora_up.weight, self.alpha_value())]
        if self.is_conv:
            b, c, h, w = hidden_states.shape
            x = hidden_states.permute(0, 2, 3, 1).reshape(b * h * w, c)
            y = F_hip.lora_linear(x, W16, Wt16, b32, site)
            y = y.view(b, h, w, -1).permute(0, 3, 1, 2)
        else:
            y = F_hip.lora_linear(hidden_states, W16, Wt16, b32, site)
        return y if y.dtype == out_dtype else y.to(out_dtype)
