"""Model construction / weight loading for the SD-1.5 stack.

`pretrained_path` follows the reference's YAML key (`models.pretrained_path`, diffusers directory layout:
unet/ vae/ text_encoder/ tokenizer/ scheduler/). When the directory exists its safetensors / .bin weights are
loaded into the local modules (state-dict keys are diffusers-compatible). There is no network and no SD-1.5
checkpoint in the build/benchmark environment, so `synthetic://<preset>?seed=N` builds seeded random-init
modules of the same architecture:  sd15 (full size), small (2 UNet levels, GPU-valid head dims), tiny (CPU tests).
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


def load_unet(path):
    preset, seed = parse_synthetic(path)
    if preset is not None:
        return _seeded(seed, lambda: UNet2DConditionModel(**PRESETS[preset]['unet']))
    m = UNet2DConditionModel()
    m.load_state_dict(_load_state(os.path.join(path, 'unet')))
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
    m = CLIPTextModel()
    m.load_state_dict(remap_text_encoder_keys(_load_state(os.path.join(path, 'text_encoder'))))
    return m


def load_vae(path):
    preset, seed = parse_synthetic(path)
    if preset is not None:
        return _seeded(seed + 2, lambda: AutoencoderKL(**PRESETS[preset]['vae']))
    m = AutoencoderKL()
    m.load_state_dict(remap_vae_keys(_load_state(os.path.join(path, 'vae'))))
    return m


def load_scheduler(path, kind='ddpm'):
    # SD-1.5 scheduler_config.json: scaled_linear 0.00085 -> 0.012, 1000 steps, epsilon prediction
    return DDPMScheduler() if kind == 'ddpm' else DPMSolverMultistepScheduler()


__all__ = ['load_unet', 'load_text_encoder', 'load_vae', 'load_scheduler', 'load_tokenizer', 'parse_synthetic']
