"""Model construction / weight loading for the SD-1.5 stack.

`pretrained_path` follows the reference's YAML key (`models.pretrained_path`, diffusers directory layout:
unet/ vae/ text_encoder/ tokenizer/ scheduler/). When the directory exists its safetensors / .bin weights are
loaded into the local modules (state-dict keys are diffusers-compatible). There is no network and no SD-1.5
checkpoint in the build/benchmark environment, so `synthetic://<preset>?seed=N` builds seeded random-init
modules of the same architecture:  sd15 (full size), small (2 UNet levels, GPU-valid head dims), tiny / tiny768 (CPU tests).
"""
import os
import re

import torch

from mixofshow.models.clip import CLIPTextModel
from mixofshow.models.schedulers import DDPMScheduler, DPMSolverMultistepScheduler
from mixofshow.models.unet_2d_condition import UNet2DConditionModel
from mixofshow.models.vae import AutoencoderKL
from mixofshow.utils.tokenizer import load_tokenizer

PRESETS = {
    'sd15': dict(unet=dict(), clip=dict(), vae=dict()),
    'small': dict(unet=dict(block_out_channels=(320, 640), layers_per_block=1), clip=dict(num_hidden_layers=2),
                  vae=dict(block_out_channels=(32, 64, 64, 64))),
    'tiny': dict(unet=dict(block_out_channels=(32, 64), layers_per_block=1, attention_head_dim=2,
                           cross_attention_dim=64, norm_num_groups=8),
                 clip=dict(hidden_size=64, num_attention_heads=2, intermediate_size=128, num_hidden_layers=1),
                 vae=dict(block_out_channels=(32, 32, 32, 32))),
    # tiny UNet / VAE, but a 768-wide text tower: the reference's gradient fusion hard-codes 768 (gradient_fusion.py:202),
    # so its own merge_* functions can run on this preset (tests/golden/make_golden.py fusion)
    'tiny768': dict(unet=dict(block_out_channels=(32, 64), layers_per_block=1, attention_head_dim=2,
                              cross_attention_dim=768, norm_num_groups=8),
                    clip=dict(hidden_size=768, num_attention_heads=12, intermediate_size=128, num_hidden_layers=1),
                    vae=dict(block_out_channels=(32, 32, 32, 32))),
}


def parse_synthetic(path):
    if path is None:
        return 'sd15', 0
    m = re.match(r'^synthetic://([a-z0-9]+)(?:\?seed=(\d+))?$', str(path))
    if m:
        return m.group(1), int(m.group(2) or 0)
    if not os.path.isdir(str(path)):
        raise FileNotFoundError(
            f'pretrained_path {path!r} is neither a diffusers model directory nor synthetic://<preset>[?seed=N]')
    return None, 0


def _load_state(folder):
    for name in ('diffusion_pytorch_model.safetensors', 'model.safetensors'):
        p = os.path.join(folder, name)
        if os.path.isfile(p):
            from safetensors.torch import load_file
            return load_file(p)
    for name in ('diffusion_pytorch_model.bin', 'pytorch_model.bin'):
        p = os.path.join(folder, name)
        if os.path.isfile(p):
            return torch.load(p, map_location='cpu')
    raise FileNotFoundError(f'no weights found in {folder}')


def _model_kwargs(folder, cls):
    """Constructor arguments from `<folder>/config.json` (the diffusers layout keeps one per sub-model; real
    checkpoints carry many more keys than the local modules take — only the ones in the constructor are used).
    Missing file = the SD-1.5 defaults."""
    import inspect
    import json
    cfg_path = os.path.join(folder, 'config.json')
    if not os.path.isfile(cfg_path):
        return {}
    with open(cfg_path) as f:
        cfg = json.load(f)
    accepted = set(inspect.signature(cls.__init__).parameters) - {'self'}
    out = {k: v for k, v in cfg.items() if k in accepted}
    if isinstance(out.get('attention_head_dim'), (list, tuple)):   # diffusers allows one value per block
        out['attention_head_dim'] = out['attention_head_dim'][0]
    for k in ('block_out_channels', ):
        if k in out:
            out[k] = tuple(out[k])
    return out


def save_model_config(model, folder):
    """`config.json` next to the weights (what `_model_kwargs` reads back)."""
    import json
    cfg = {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(model.config).items()}
    with open(os.path.join(folder, 'config.json'), 'w') as f:
        json.dump(cfg, f, indent=1)


def remap_text_encoder_keys(sd):
    """Accept both key styles: reference-era `text_model.encoder...` and transformers>=5 `encoder...`."""
    out = {}
    for k, v in sd.items():
        if k.endswith('position_ids'):
            continue
        if not k.startswith('text_model.'):
            k = 'text_model.' + k
        out[k] = v
    return out


def remap_vae_keys(sd):
    """diffusers < 0.18 named the VAE attention projections query/key/value/proj_attn."""
    ren = {'.query.': '.to_q.', '.key.': '.to_k.', '.value.': '.to_v.', '.proj_attn.': '.to_out.0.'}
    out = {}
    for k, v in sd.items():
        for a, b in ren.items():
            k = k.replace(a, b)
        if k.endswith(('to_q.weight', 'to_k.weight', 'to_v.weight', 'to_out.0.weight')) and v.dim() == 4:
            v = v[:, :, 0, 0]
        out[k] = v
    return out


def _seeded(seed, fn):
    devices = []
    with torch.random.fork_rng(devices=devices):
        torch.manual_seed(seed)
        return fn()


@torch.no_grad()
def calibrate_synthetic_unet(unet, branch_gain=0.35, out_gain=0.5, qk_gain=3.0, attn_out_gain=1.0):
    """Make a random-init UNet behave like a trained epsilon-predictor as far as SCALES go (synthetic
    weights "scaled so activations stay O(1)"), so that 50 sampler steps are a contraction instead of a chaotic
    amplifier (an untrained net predicts eps ~ 0 and DPM-Solver then multiplies the latent by 1/alpha_T ~ 15):

      * every residual branch ends in a down-scaled layer (ResnetBlock2D.conv2, Attention.to_out[0], FeedForward.net[2],
        Transformer2DModel.proj_out x `branch_gain`, zero bias) — original LDM zero-initialises exactly these;
      * an identity path carries the input latent to the output: conv_in writes +-x_c into channels 2c / 2c+1, the
        outermost skip connection hands them to the last up-block ResNet whose 1x1 shortcut copies them, and conv_out
        reads SiLU(u) - SiLU(-u) = u back, so eps_hat = x / rms(x) + (everything else) * `out_gain`: at high noise that
        IS the noise, and the sampler's x0-prediction stays O(1) for every t.

    All attention / LoRA / text paths still feed eps_hat through the second term; tests report the sensitivity."""
    from mixofshow.models.unet_2d_condition import FeedForward, ResnetBlock2D, Transformer2DModel
    from mixofshow.models.attention import Attention
    for m in unet.modules():
        outs = []
        if isinstance(m, ResnetBlock2D):
            outs.append(m.conv2)
        elif isinstance(m, Attention):
            # default init gives attention logits of std ~0.3 (softmax ~ uniform: the layer degenerates to a mean of V
            # and its output hardly depends on the scores); trained SD-1.5 logits are several units wide
            m.to_q.weight.mul_(qk_gain)
            m.to_k.weight.mul_(qk_gain)
            m.to_out[0].weight.mul_(attn_out_gain / branch_gain)
            outs.append(m.to_out[0])
        elif isinstance(m, FeedForward):
            outs.append(m.net[2])
        elif isinstance(m, Transformer2DModel):
            outs.append(m.proj_out)
        for layer in outs:
            layer.weight.mul_(branch_gain)
            if layer.bias is not None:
                layer.bias.zero_()
    cin = unet.conv_in.weight.shape[1]
    n_id = 2 * cin
    last = unet.up_blocks[-1].resnets[-1]
    c0 = unet.conv_in.weight.shape[0]
    assert last.conv_shortcut is not None and last.conv_shortcut.weight.shape[1] == 2 * c0 and n_id <= c0
    unet.conv_in.weight[:n_id].zero_()
    unet.conv_in.bias[:n_id].zero_()
    last.conv_shortcut.weight[:n_id].zero_()
    last.conv_shortcut.bias[:n_id].zero_()
    unet.conv_out.weight.mul_(out_gain)
    unet.conv_out.bias.zero_()
    unet.conv_out.weight[:, :n_id].zero_()
    kh = unet.conv_in.weight.shape[2] // 2
    for c in range(cin):
        unet.conv_in.weight[2 * c, c, kh, kh] = 1.0
        unet.conv_in.weight[2 * c + 1, c, kh, kh] = -1.0
        last.conv_shortcut.weight[2 * c, c0 + 2 * c, 0, 0] = 1.0          # skip half of cat([x, skip]) = [c0, 2 c0)
        last.conv_shortcut.weight[2 * c + 1, c0 + 2 * c + 1, 0, 0] = 1.0
        if c < unet.conv_out.weight.shape[0]:
            unet.conv_out.weight[c, 2 * c, kh, kh] = 1.0
            unet.conv_out.weight[c, 2 * c + 1, kh, kh] = -1.0
    return unet


def load_unet(path):
    preset, seed = parse_synthetic(path)
    if preset is not None:
        return _seeded(seed, lambda: calibrate_synthetic_unet(UNet2DConditionModel(**PRESETS[preset]['unet'])))
    folder = os.path.join(path, 'unet')
    m = UNet2DConditionModel(**_model_kwargs(folder, UNet2DConditionModel))
    m.load_state_dict(_load_state(folder))
    return m


def load_text_encoder(path):
    preset, seed = parse_synthetic(path)
    if preset is not None:
        def make():
            m = CLIPTextModel(**PRESETS[preset]['clip'])
            with torch.no_grad():
                # real CLIP token embeddings have row norms ~0.4 (the emb_norm_threshold 0.55 of the YAMLs assumes
                # that scale); nn.Embedding's N(0,1) init would give norms ~27 and freeze the concept rows at step 1
                m.text_model.embeddings.token_embedding.weight.mul_(0.014)
                m.text_model.embeddings.position_embedding.weight.mul_(0.014)
            return m
        return _seeded(seed + 1, make)
    folder = os.path.join(path, 'text_encoder')
    sd = remap_text_encoder_keys(_load_state(folder))
    kw = _model_kwargs(folder, CLIPTextModel)
    # a fused / tuned checkpoint carries the concept rows: the table size is whatever was saved
    kw['vocab_size'] = sd['text_model.embeddings.token_embedding.weight'].shape[0]
    m = CLIPTextModel(**kw)
    m.load_state_dict(sd)
    return m


def load_vae(path):
    preset, seed = parse_synthetic(path)
    if preset is not None:
        return _seeded(seed + 2, lambda: AutoencoderKL(**PRESETS[preset]['vae']))
    folder = os.path.join(path, 'vae')
    m = AutoencoderKL(**_model_kwargs(folder, AutoencoderKL))
    m.load_state_dict(remap_vae_keys(_load_state(folder)))
    return m


def load_scheduler(path, kind='ddpm'):
    # SD-1.5 scheduler_config.json: scaled_linear 0.00085 -> 0.012, 1000 steps, epsilon prediction
    return DDPMScheduler() if kind == 'ddpm' else DPMSolverMultistepScheduler()


__all__ = ['load_unet', 'load_text_encoder', 'load_vae', 'load_scheduler', 'load_tokenizer', 'parse_synthetic',
           'save_model_config']
