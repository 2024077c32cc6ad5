"""AutoencoderKL (SD-1.5 VAE) — plumbing around the hot path: images -> latents for the training step
(reference trainer_edlora.py:203-204) and latents -> images for validation. Plain PyTorch-ROCm modules with
diffusers-compatible state-dict keys (0.19 naming: `mid_block.attentions.0.to_q` ...). Its single-head d=512 mid-block attention runs as
scores GEMM -> row softmax -> values GEMM on the library kernels when no gradient is needed (training: frozen encoder),
torch SDPA otherwise."""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from mixofshow.hip.functional import group_norm_act
from mixofshow.models.unet_2d_condition import Downsample2D, ResnetBlock2D, Upsample2D


class VaeAttention(nn.Module):

    def __init__(self, channels, groups=32):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        # token-major view: free for channels_last activations, one copy for NCHW
        y = group_norm_act(self.group_norm, x, False).permute(0, 2, 3, 1).reshape(b, h * w, c)
        q, k, v = self.to_q(y), self.to_k(y), self.to_v(y)
        half = q.is_cuda and q.dtype in (torch.float16, torch.bfloat16)
        no_grad = not (torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad))
        if half and no_grad and c % 8 == 0 and (h * w) % 8 == 0 and h * w <= 32768:
            from mixofshow.hip import ops            # frozen VAE (training encodes under no grad): library kernels
            o = ops.single_head_attention_nograd(q.contiguous(), k.contiguous(), v.contiguous(), c**-0.5)
        else:
            o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = self.to_out[0](o)
        return o.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


class _Mid(nn.Module):

    def __init__(self, c):
        super().__init__()
        self.attentions = nn.ModuleList([VaeAttention(c)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, eps=1e-6), ResnetBlock2D(c, c, None, eps=1e-6)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _EncBlock(nn.Module):

    def __init__(self, cin, cout, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6) for i in range(2)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.downsamplers[0](x) if self.downsamplers is not None else x


class _DecBlock(nn.Module):

    def __init__(self, cin, cout, up):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, eps=1e-6) for i in range(3)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return self.upsamplers[0](x) if self.upsamplers is not None else x


class Encoder(nn.Module):

    def __init__(self, ch=(128, 256, 512, 512), in_channels=3, latent=4):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        blocks, c = [], ch[0]
        for i, co in enumerate(ch):
            blocks.append(_EncBlock(c, co, i != len(ch) - 1))
            c = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid(c)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(c, 2 * latent, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        return self.conv_out(group_norm_act(self.conv_norm_out, self.mid_block(x), True))


class Decoder(nn.Module):

    def __init__(self, ch=(128, 256, 512, 512), out_channels=3, latent=4):
        super().__init__()
        rev = ch[::-1]
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0])
        blocks, c = [], rev[0]
        for i, co in enumerate(rev):
            blocks.append(_DecBlock(c, co, i != len(rev) - 1))
            c = co
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(32, c, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(group_norm_act(self.conv_norm_out, x, True))


class DiagonalGaussianDistribution:

    def __init__(self, parameters):
        self.mean, logvar = parameters.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None, noise=None):
        """`noise`: pre-drawn standard-normal tensor (device-independent randomness for parity tests / multi-rank runs)."""
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise.to(self.mean)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):

    def __init__(self, block_out_channels=(128, 256, 512, 512), latent_channels=4):
        super().__init__()
        self.config = SimpleNamespace(block_out_channels=tuple(block_out_channels), latent_channels=latent_channels,
                                      scaling_factor=0.18215)
        self.encoder = Encoder(block_out_channels, 3, latent_channels)
        self.decoder = Decoder(block_out_channels, 3, latent_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))
