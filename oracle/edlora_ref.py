"""ORACLE — test infrastructure, NOT product code.

CPU restatement (plain torch, any float dtype; run it in fp32/fp64 for ground truth) of the reference's
ED-LoRA hot path. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (mix-of-show_amd/mixofshow) never does.

Pinned against the REAL reference code: tests/golden/make_golden.py imports /root/reference with stubbed
third-party modules and records its outputs; tests/test_oracle_golden.py checks this file against them.
What stays "from memory" is the diffusers `Attention` helper semantics (oracle/attention_shim.py).

Each function cites the reference lines it follows (paths relative to TencentARC/Mix-of-Show).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- mixofshow/models/edlora.py:221-246 -------------------------------------------------------
class LoRALinearLayerRef(nn.Module):
    """y = orig(x) + alpha * up(down(x)); down ~ kaiming_uniform(a=sqrt 5), up = 0; alpha is a buffer;
    patches original_module.forward in place (:241-242)."""

    def __init__(self, name, original_module, rank=4, alpha=1):
        super().__init__()
        self.name = name
        if isinstance(original_module, nn.Conv2d):
            self.lora_down = nn.Conv2d(original_module.in_channels, rank, (1, 1), bias=False)
            self.lora_up = nn.Conv2d(rank, original_module.out_channels, (1, 1), bias=False)
        else:
            self.lora_down = nn.Linear(original_module.in_features, rank, bias=False)
            self.lora_up = nn.Linear(rank, original_module.out_features, bias=False)
        self.register_buffer('alpha', torch.tensor(alpha))
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)
        self.original_forward = original_module.forward
        original_module.forward = self.forward

    def forward(self, x):
        return self.original_forward(x) + self.alpha * self.lora_up(self.lora_down(x))


def lora_linear_ref(x, W, bias, down, up, alpha):
    """Functional form of the above for a Linear site."""
    return F.linear(x, W, bias) + alpha * F.linear(F.linear(x, down), up)


# ---- mixofshow/models/edlora.py:103-173 (baddbmm branch :155-156) and :22-100 ---------------------
def scores_times_value_ref(attn, q, k, v, limit_bytes=2**33):
    """`torch.bmm(attn.get_attention_scores(q, k), v)` (:81-83). The reference materialises the whole (B*H, N, M) probability
    tensor; at the reference's shipped 1024x2048 example (N = 32768 self-attention keys) that is 69 GB in fp32, so above
    `limit_bytes` the SAME two calls run on slices of the leading (batch x head) dimension -- every slice is independent of
    the others in both calls, the result is identical."""
    n = q.shape[0]
    per = q.shape[1] * k.shape[1] * 4
    step = max(1, min(n, limit_bytes // max(per, 1)))
    if step >= n:
        return torch.bmm(attn.get_attention_scores(q, k, None), v)
    return torch.cat([torch.bmm(attn.get_attention_scores(q[i:i + step], k[i:i + step], None), v[i:i + step])
                      for i in range(0, n, step)])


def _attention_layer_ref(attn, hidden_states, encoder_hidden_states, layer_idx, controller=None, place=None):
    residual = hidden_states
    if attn.spatial_norm is not None:
        hidden_states = attn.spatial_norm(hidden_states, None)
    ndim = hidden_states.ndim
    if ndim == 4:
        b, c, h, w = hidden_states.shape
        hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
    is_cross = encoder_hidden_states is not None
    if not is_cross:
        context = hidden_states
    elif encoder_hidden_states.dim() == 4:          # multi-layer embedding (:56-57 / :130-131)
        context = encoder_hidden_states[:, layer_idx, ...]
    else:
        context = encoder_hidden_states
    assert not attn.norm_cross
    attn.prepare_attention_mask(None, context.shape[1], context.shape[0])
    if attn.group_norm is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    q = attn.head_to_batch_dim(attn.to_q(hidden_states)).contiguous()
    k = attn.head_to_batch_dim(attn.to_k(context)).contiguous()
    v = attn.head_to_batch_dim(attn.to_v(context)).contiguous()
    if controller is not None:
        probs = attn.get_attention_scores(q, k, None)
        probs = controller(probs, is_cross, place)  # :82 — the tensor keeps its autograd graph
        out = attn.batch_to_head_dim(torch.bmm(probs, v))
    else:
        out = attn.batch_to_head_dim(scores_times_value_ref(attn, q, k, v))
    out = attn.to_out[1](attn.to_out[0](out))
    if ndim == 4:
        out = out.transpose(-1, -2).reshape(b, c, h, w)
    if attn.residual_connection:
        out = out + residual
    return out / attn.rescale_output_factor


class EDLoRA_AttnProcessorRef:

    def __init__(self, cross_attention_idx):
        self.cross_attention_idx = cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        return _attention_layer_ref(attn, hidden_states, encoder_hidden_states, self.cross_attention_idx)


class EDLoRA_Control_AttnProcessorRef:

    def __init__(self, cross_attention_idx, place_in_unet, controller):
        self.cross_attention_idx, self.place_in_unet, self.controller = cross_attention_idx, place_in_unet, controller

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        return _attention_layer_ref(attn, hidden_states, encoder_hidden_states, self.cross_attention_idx,
                                    controller=self.controller, place=self.place_in_unet)


def install_ref_processors(unet, controller=None, control=False):
    """Processor installation order of edlora.py:176-218 (down -> mid -> up, attn2 only)."""

    def visit(module, count, place):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention' and 'attn2' in name:
                layer.set_processor(EDLoRA_Control_AttnProcessorRef(count, place, controller) if control else
                                    EDLoRA_AttnProcessorRef(count))
                count += 1
            else:
                count = visit(layer, count, place)
        return count

    n = visit(unet.down_blocks, 0, 'down')
    n = visit(unet.mid_block, n, 'mid')
    n = visit(unet.up_blocks, n, 'up')
    if controller is not None:
        controller.num_att_layers = n
    return n


class PlainAttnProcessorRef:
    """diffusers' default processor for attn1 (self-attention), same math as the branch above."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        return _attention_layer_ref(attn, hidden_states, encoder_hidden_states, 0)


# ---- mixofshow/utils/ptp_util.py:22-108 -----------------------------------------------------------
class AttentionStoreRef:
    """Full-map store: keeps every (B*H, N, 77) probability tensor of a UNet forward."""

    def __init__(self, training=True):
        self.training = training
        self.num_att_layers = -1
        self.reset()

    def reset(self):
        self.cur_step, self.cur_att_layer = 0, 0
        self.step_store = self._empty()
        self.attention_store = {}

    @staticmethod
    def _empty():
        return {k: [] for k in ('down_cross', 'mid_cross', 'up_cross', 'down_self', 'mid_self', 'up_self')}

    def __call__(self, attn, is_cross, place):
        key = f"{place}_{'cross' if is_cross else 'self'}"
        if self.training:
            self.step_store[key].append(attn)
        else:                                        # :45-46: only the conditional half
            half = attn.shape[0] // 2
            self.step_store[key].append(attn[half:])
        self.cur_att_layer += 1
        if self.cur_att_layer == self.num_att_layers:
            self.cur_att_layer = 0
            self.cur_step += 1
            if not self.attention_store:             # between_steps :84-91
                self.attention_store = self.step_store
            else:
                for k in self.attention_store:
                    for i in range(len(self.attention_store[k])):
                        self.attention_store[k][i] = self.attention_store[k][i] + self.step_store[k][i]
            self.step_store = self._empty()
        return attn

    def get_average_attention(self):
        return {k: [x / self.cur_step for x in v] for k, v in self.attention_store.items()}


# ---- mixofshow/pipelines/trainer_edlora.py:263-313 --------------------------------------------------
def cal_attn_reg_ref(attention_maps, masks, text_input_ids, concept_token_ids, attn_reg_weight, reg_full_identity,
                     strict_resolutions=True):
    """attention_maps: {'down_cross': [(B*H, N, 77), ...], ...}; masks (B,1,64,64); ids (B*16, 77)."""
    B = masks.shape[0]
    ids = text_input_ids.reshape(B, -1, text_input_ids.shape[-1])
    pos = []
    for row in ids:                                  # positions from the FIRST of the 16 prompts (:278-279)
        first = row[0].tolist()
        pos.append([i for i, t in enumerate(first) if t in concept_token_ids])
    groups = {'64': [], '32': [], '16': [], '8': []}
    for maps in attention_maps.values():
        for m in maps:
            res = int(math.sqrt(m.shape[1]))
            groups[str(res)].append(m.reshape(B, -1, res, res, m.shape[-1]))
    total = 0
    for res, maps in groups.items():
        if not maps and not strict_resolutions:      # the reference itself needs all 4 groups (512x512 inputs only)
            continue
        cm = torch.cat(maps, dim=-4)                 # concat heads of all layers at this resolution
        cm = cm.sum(-4) / cm.shape[-4]
        cm = torch.stack([bm[..., p] for p, bm in zip(pos, cm)])
        adj, subj = cm[..., 0], cm[..., 1]
        subj = subj / subj.max()                     # global max over the batch (:300-301)
        adj = adj / adj.max()
        gt = F.interpolate(masks, size=subj.shape[1:], mode='nearest').squeeze(1)
        if reg_full_identity:
            l_subj = F.mse_loss(subj.float(), gt.float(), reduction='mean')
        else:
            l_subj = subj[gt == 0].mean()
        l_adj = adj[gt == 0].mean()
        total = total + attn_reg_weight * (l_subj + l_adj)
    return total


# ---- mixofshow/pipelines/trainer_edlora.py:247-252 --------------------------------------------------
def masked_mse_ref(model_pred, target, loss_mask):
    loss = F.mse_loss(model_pred.float(), target.float(), reduction='none')
    return ((loss * loss_mask).sum([1, 2, 3]) / loss_mask.sum([1, 2, 3])).mean()


# ---- mixofshow/pipelines/pipeline_edlora.py:18-29 -----------------------------------------------------
def bind_concept_prompt_ref(prompts, new_concept_cfg):
    if isinstance(prompts, str):
        prompts = [prompts]
    out = []
    for prompt in prompts:
        per_layer = [prompt] * 16
        for concept, cfg in new_concept_cfg.items():
            per_layer = [p.replace(concept, tok) for p, tok in zip(per_layer, cfg['concept_token_names'])]
        out.extend(per_layer)
    return out


# ---- mixofshow/utils/convert_edlora_to_diffusers.py:33-76 & gradient_fusion.py:99-143 ---------------------
_TE_SITES = ('q_proj', 'k_proj', 'v_proj', 'out_proj', 'fc1', 'fc2')
_UNET_SITES = ('to_q', 'to_k', 'to_v', 'to_out.0', 'ff.net.0.proj', 'ff.net.2', 'proj_out', 'proj_in')


def lora_down_key(weight_key, model_type):
    sites = _TE_SITES if model_type == 'text_encoder' else _UNET_SITES
    for s in sites:
        weight_key = weight_key.replace(f'{s}.weight', f'{s}.lora_down.weight')
    return weight_key


def merge_lora_into_weight_ref(original_state_dict, lora_state_dict, model_type, alpha, layer_names=None):
    """W' = W + alpha * up @ down for every key that has a LoRA pair (4-D: squeeze/unsqueeze)."""
    new_sd = {k: v.clone() for k, v in original_state_dict.items()}
    keys = list(new_sd.keys()) if layer_names is None else list(layer_names)
    n = 0
    for k in keys:
        dn = lora_down_key(k, model_type)
        up = dn.replace('lora_down', 'lora_up')
        if up in lora_state_dict:
            n += 1
            W = new_sd[k]
            d, u = lora_state_dict[dn].to(W.device), lora_state_dict[up].to(W.device)
            delta = (u.squeeze() @ d.squeeze())[..., None, None] if W.dim() == 4 else u @ d
            new_sd[k] = W + alpha * delta
    return new_sd, n
