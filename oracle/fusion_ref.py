"""ORACLE — test infrastructure. CPU restatement of gradient fusion: the feature collection (hooks, analytic cross-K/V
targets, text-encoder and spatial-layer passes; second half of this file) and the least squares of the reference
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


# =====================================================================================================
# Feature collection of gradient fusion (gradient_fusion.py:150-167 hook, :183-214 text features,
# :325-457 cross-attention K/V, :460-576 text encoder, :579-624 sampling loop, :627-747 spatial layers).
#
# Restated the way the reference does it: forward hooks on nn.Linear / 1x1 Conv2d modules append (input,
# bias-free output) to host lists, one list pair per module; after every concept the lists are concatenated;
# each layer is then solved with update_quasi_newton_ref on the STORED features. The models are whatever the
# caller passes (tests: the product's module classes running the ORACLE processors, so every projection is a
# real nn.Linear call and the hooks fire exactly as in the reference).
# =====================================================================================================
TEMPLATE_SIMPLE_REF = 'photo of a {}'


class FeatureStoreRef:
    """module_io_recoder + record_feature globals of the reference (:146-167) as an object."""

    def __init__(self):
        self.store = {}
        self.record = False

    def hook_for(self, module_name):
        def hook(module, feature_in, feature_out):
            ent = self.store.setdefault(module_name, {'input': [], 'output': []})
            if not self.record:
                return
            ent['input'].append(feature_in[0].detach().cpu())
            if module.bias is not None:                                   # remove bias (:156-162)
                bias = module.bias[:, None, None] if feature_out.dim() == 4 else module.bias
                ent['output'].append((feature_out - bias).detach().cpu())
            else:
                ent['output'].append(feature_out.detach().cpu())
        return hook


@torch.no_grad()
def get_text_feature_ref(prompts, tokenizer, text_encoder, device, return_type='category_embedding'):
    """:183-214 — per prompt, un-padded; rows of tokens with id >= 49407 (new tokens and the end token)."""
    if return_type == 'category_embedding':
        feats = []
        for text in prompts:
            tokens = tokenizer(text, truncation=True, max_length=tokenizer.model_max_length, padding='do_not_pad').input_ids
            pos = torch.where(torch.tensor(tokens) >= 49407)[0]
            out = text_encoder(torch.LongTensor(tokens).reshape(1, -1).to(device))[0][:, pos]
            feats.append(out.reshape(-1, out.shape[-1]))
        return torch.cat(feats, 0).float()
    ids = tokenizer(prompts, padding='max_length', max_length=tokenizer.model_max_length, truncation=True,
                    return_tensors='pt').input_ids
    return text_encoder(ids.to(device))[0]


def _lora_keys(lora_list):
    keys = set()
    for lora in lora_list:
        keys |= {k.replace('.lora_down', '').replace('.lora_up', '') for k in lora.keys()}
    return keys


def merge_text_encoder_ref(concept_list, iters, new_concept_cfg, tokenizer, text_encoder, text_encoder_list, device,
                           bind, solve=update_quasi_newton_ref, return_features=False):
    """:460-576. `bind` = bind_concept_prompt (oracle or reference)."""
    from oracle.edlora_ref import merge_lora_into_weight_ref
    layer_names = _lora_keys(text_encoder_list)
    cand = [n for n in ('q_proj', 'k_proj', 'v_proj', 'out_proj', 'fc1', 'fc2') if any(n in k for k in layer_names)]
    fs = FeatureStoreRef()
    handles = [m.register_forward_hook(fs.hook_for(name)) for name, m in text_encoder.named_modules()
               if any(c in name for c in cand)]
    original = {k: v.detach().clone() for k, v in text_encoder.state_dict().items()}
    X, Y = {}, {}
    for concept, lora in zip(concept_list, text_encoder_list):
        merged, _ = merge_lora_into_weight_ref(original, {k: v.to(device) for k, v in lora.items()}, 'text_encoder',
                                               concept['text_encoder_alpha'], layer_names=layer_names)
        text_encoder.load_state_dict(merged)
        prompts = bind([TEMPLATE_SIMPLE_REF.format(concept['concept_name']), concept['concept_name']], new_concept_cfg)
        fs.store, fs.record = {}, True
        get_text_feature_ref(prompts, tokenizer, text_encoder, device)
        fs.record = False
        for ln in layer_names:
            ent = fs.store[ln.replace('.weight', '')]
            X.setdefault(ln, []).append(torch.cat([f.reshape(-1, f.shape[-1]) for f in ent['input']], 0))
            Y.setdefault(ln, []).append(torch.cat([f.reshape(-1, f.shape[-1]) for f in ent['output']], 0))
    for h in handles:
        h.remove()
    text_encoder.load_state_dict(original)
    X = {k: torch.cat(v, 0) for k, v in X.items()}
    Y = {k: torch.cat(v, 0) for k, v in Y.items()}
    if return_features:
        return X, Y, original
    return {ln: solve(X[ln].float(), Y[ln].float(), original[ln].float().cpu().clone(), iters) for ln in sorted(layer_names)}


def cross_kv_layer_names_ref(unet):
    """:333-369 — (cross-attention index, weight name) in down -> mid -> up order, to_k then to_v."""
    names, idx = [], -1
    for prefix, block in (('down_blocks.', unet.down_blocks), ('mid_block.', unet.mid_block), ('up_blocks.', unet.up_blocks)):
        for name, _ in block.named_parameters():
            if 'attn2.to_k' in name:
                idx += 1
                names.append((idx, prefix + name))
                names.append((idx, prefix + name.replace('to_k', 'to_v')))
    return names


def merge_kv_in_cross_attention_ref(concept_list, iters, new_concept_cfg, tokenizer, text_encoder, unet, crosskv_list,
                                    device, bind, solve=update_quasi_newton_ref, return_features=False):
    """:325-457 — targets are analytic: (W + alpha * up @ down) applied to the concept-token text features."""
    names = cross_kv_layer_names_ref(unet)
    sd = unet.state_dict()
    X, Y = {}, {}
    for concept, tuned in zip(concept_list, crosskv_list):
        prompts = bind([TEMPLATE_SIMPLE_REF.format(concept['concept_name']), concept['concept_name']], new_concept_cfg)
        n = len(prompts) // 16
        layer_prompts = [tuple(prompts[j * 16 + i] for j in range(n)) for i in range(16)]
        for layer_idx, ln in names:
            dn = ln.replace('to_k.weight', 'to_k.lora_down.weight').replace('to_v.weight', 'to_v.lora_down.weight')
            up = dn.replace('lora_down', 'lora_up')
            merged = sd[ln] + concept['unet_alpha'] * tuned[up].to(device) @ tuned[dn].to(device)   # promotes to fp32
            feat = get_text_feature_ref(list(layer_prompts[layer_idx]), tokenizer, text_encoder, device).cpu()
            X.setdefault(ln, []).append(feat)
            Y.setdefault(ln, []).append((merged.cpu() @ feat.T).T)
    X = {k: torch.cat(v, 0) for k, v in X.items()}
    Y = {k: torch.cat(v, 0) for k, v in Y.items()}
    if return_features:
        return X, Y, sd
    return {ln: solve(X[ln].float(), Y[ln].float(), sd[ln].float().cpu().clone(), iters) for _, ln in names}


@torch.no_grad()
def decode_to_latents_ref(prompt, new_concept_cfg, tokenizer, text_encoder, unet, scheduler, steps, device, record_nums,
                          batch_size, fs, bind):
    """:579-624 — 512x512, no CFG; latents from the GLOBAL CPU generator (callers seed it)."""
    prompts = bind([prompt], new_concept_cfg)
    emb = get_text_feature_ref(prompts, tokenizer, text_encoder, device, return_type='full_embedding').unsqueeze(0)
    emb = emb.repeat((batch_size, 1, 1, 1))
    latents = torch.randn((batch_size, unet.in_channels, 64, 64)).to(device, dtype=emb.dtype)
    scheduler.set_timesteps(steps)
    latents = latents * scheduler.init_noise_sigma
    ts = scheduler.timesteps
    step = ts.size(0) // record_nums
    rec_ts = ts[torch.arange(0, ts.size(0), step=step)[:record_nums]]
    for t in ts:
        fs.record = bool((rec_ts == t).any())
        noise_pred = unet(scheduler.scale_model_input(latents, t), t, encoder_hidden_states=emb).sample
        latents = scheduler.step(noise_pred, t, latents).prev_sample
    fs.record = False
    return latents, emb


def merge_spatial_attention_ref(concept_list, iters, new_concept_cfg, tokenizer, text_encoder, unet, spatial_list,
                                scheduler, device, bind, solve=update_quasi_newton_ref, return_features=False,
                                steps=20, record_nums=20):
    """:627-747. The caller installs the oracle's EDLoRA processors (the reference calls
    revise_edlora_unet_attention_forward here, :661)."""
    from oracle.edlora_ref import install_ref_processors, merge_lora_into_weight_ref
    layer_names = _lora_keys(spatial_list)
    cand = [n for n in ('attn2.to_q', 'attn2.to_out.0', 'attn1.to_q', 'attn1.to_k', 'attn1.to_v', 'attn1.to_out.0',
                        'ff.net.2', 'ff.net.0.proj', 'proj_out', 'proj_in') if any(n in k for k in layer_names)]
    fs = FeatureStoreRef()
    handles = [m.register_forward_hook(fs.hook_for(name)) for name, m in unet.named_modules()
               if any(c in name for c in cand)]
    original = {k: v.detach().clone() for k, v in unet.state_dict().items()}
    install_ref_processors(unet)
    X, Y = {}, {}
    for concept, tuned in zip(concept_list, spatial_list):
        fs.store = {}
        merged, _ = merge_lora_into_weight_ref(original, {k: v.to(device) for k, v in tuned.items()}, 'unet',
                                               concept['unet_alpha'], layer_names=layer_names)
        unet.load_state_dict(merged)
        decode_to_latents_ref(TEMPLATE_SIMPLE_REF.format(concept['concept_name']), new_concept_cfg, tokenizer,
                              text_encoder, unet, scheduler, steps, device, record_nums, 1, fs, bind)
        for ln in layer_names:
            ent = fs.store[ln.replace('.weight', '')]
            X.setdefault(ln, []).append(torch.cat(ent['input'], 0))
            Y.setdefault(ln, []).append(torch.cat(ent['output'], 0))
    for h in handles:
        h.remove()
    unet.load_state_dict(original)
    X = {k: torch.cat(v, 0) for k, v in X.items()}
    Y = {k: torch.cat(v, 0) for k, v in Y.items()}

    def rows(t):
        return t if t.dim() == 4 else t.reshape(-1, t.shape[-1])

    if return_features:
        return {k: rows(v) for k, v in X.items()}, {k: rows(v) for k, v in Y.items()}, original
    return {ln: solve(rows(X[ln]).float(), rows(Y[ln]).float(), original[ln].float().cpu().clone(), iters)
            for ln in sorted(layer_names)}
