"""ORACLE — test infrastructure. CPU restatement of the regional attention of the reference
(mixofshow/pipelines/pipeline_regionally_t2iadapter.py:27-163) and of the region string grammar
(regionally_controlable_sampling.py:67-94)."""
import math

import torch


def region_boxes_ref(region_fracs, feat_h, feat_w):
    """ceil on starts, floor on ends of FRACTIONAL coordinates times the feature size (:38-39, :68-69)."""
    boxes = []
    for sh, sw, eh, ew in region_fracs:
        boxes.append((math.ceil(sh * feat_h), math.ceil(sw * feat_w), math.floor(eh * feat_h), math.floor(ew * feat_w)))
    return boxes


def region_rewrite_ref(attn, hidden_states, query, region_list, height, width):
    """:32-86 — hidden_states/query are head-batched (B*H, N, d); region_list = [(K_r, V_r, frac_box)]."""
    dtype = query.dtype
    n_tok = query.shape[1]
    down = math.sqrt(height * width / n_tok)
    fh, fw = int(height // down), int(width // down)
    boxes = region_boxes_ref([r[-1] for r in region_list], fh, fw)
    count = torch.zeros((fh, fw))
    for h0, w0, h1, w1 in boxes:
        count[h0:h1, w0:w1] += 1
    q = query.reshape(query.shape[0], fh, fw, -1)
    base = hidden_states.reshape(hidden_states.shape[0], fh, fw, -1)
    out = torch.zeros_like(base)
    out[:, count == 0, :] = base[:, count == 0, :]
    replace_ratio = 1.0                               # hard-coded (:57)
    out[:, count != 0, :] = (1 - replace_ratio) * base[:, count != 0, :]
    for (key_r, val_r, _), (h0, w0, h1, w1) in zip(region_list, boxes):
        if attn.upcast_attention:
            q, key_r = q.float(), key_r.float()
        scores = torch.einsum('bhwc,bnc->bhwn', q[:, h0:h1, w0:w1, :], key_r) * attn.scale
        if attn.upcast_softmax:
            scores = scores.float()
        probs = scores.softmax(dim=-1).to(dtype)
        upd = torch.einsum('bhwn,bnc->bhwc', probs, val_r)
        out[:, h0:h1, w0:w1, :] += replace_ratio * (upd / count.reshape(1, fh, fw, 1)[:, h0:h1, w0:w1, :].to(q.device))
    return out.reshape(out.shape[0], fh * fw, -1)


class RegionT2I_AttnProcessorRef:
    """:88-145 — registered on EVERY Attention (attn1 and attn2); region kwargs only used when cross."""

    def __init__(self, cross_attention_idx):
        self.cross_attention_idx = cross_attention_idx

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 **cross_attention_kwargs):
        attn.prepare_attention_mask(attention_mask, hidden_states.shape[1], hidden_states.shape[0])
        query = attn.to_q(hidden_states)
        is_cross = encoder_hidden_states is not None
        if not is_cross:
            context = hidden_states
        elif encoder_hidden_states.dim() == 4:
            context = encoder_hidden_states[:, self.cross_attention_idx, ...]
        else:
            context = encoder_hidden_states
        key, value = attn.to_k(context), attn.to_v(context)
        query, key, value = (attn.head_to_batch_dim(t) for t in (query, key, value))
        from oracle.edlora_ref import scores_times_value_ref
        out = scores_times_value_ref(attn, query, key, value)      # get_attention_scores + bmm (:111-116), sliced when huge
        if is_cross:
            regions = []
            for states, box in cross_attention_kwargs['region_list']:   # KeyError if absent, like the reference
                s = states[:, self.cross_attention_idx, ...] if states.dim() == 4 else states
                regions.append((attn.head_to_batch_dim(attn.to_k(s)), attn.head_to_batch_dim(attn.to_v(s)), box))
            out = region_rewrite_ref(attn, out, query, regions, cross_attention_kwargs['height'],
                                     cross_attention_kwargs['width'])
        out = attn.batch_to_head_dim(out)
        return attn.to_out[1](attn.to_out[0](out))


def install_region_processors_ref(unet):
    """:148-163 — all Attention layers get the processor; the index advances on attn2 only."""

    def visit(module, count):
        for name, layer in module.named_children():
            if layer.__class__.__name__ == 'Attention':
                layer.set_processor(RegionT2I_AttnProcessorRef(count))
                if 'attn2' in name:
                    count += 1
            else:
                count = visit(layer, count)
        return count

    n = visit(unet.down_blocks, 0)
    n = visit(unet.mid_block, n)
    return visit(unet.up_blocks, n)


def prepare_text_ref(prompt, region_prompts, height, width):
    """regionally_controlable_sampling.py:67-94: '[prompt]-*-[neg]-*-[h0, w0, h1, w1]|...' -> fractions."""
    regions = []
    for region in region_prompts.split('|'):
        if region == '':
            break
        p, n, pos = region.split('-*-')
        p, n = p.replace('[', '').replace(']', ''), n.replace('[', '').replace(']', '')
        pos = eval(pos)  # noqa: S307 — the reference evals the box literal
        if len(pos) == 0:
            pos = [0, 0, 1, 1]
        else:
            pos[0], pos[2] = pos[0] / height, pos[2] / height
            pos[1], pos[3] = pos[1] / width, pos[3] / width
        regions.append((p, n, pos))
    return (prompt, regions)
