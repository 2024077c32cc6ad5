"""(f3) "Real SD-1.5 / ChilloutMix weights load": the local UNet / VAE / CLIP modules must expose exactly the state-dict
keys and shapes of the published diffusers SD-1.5 checkpoint. No checkpoint exists offline, so the published facts the
layout is pinned to are: the per-module parameter TOTALS of runwayml/stable-diffusion-v1-5 (UNet 859,520,964; VAE
83,653,863; CLIP ViT-L/14 text encoder 123,060,480), the tensor counts, and a spot list of (key, shape) pairs spanning
every block type (taken from the diffusers 0.19 checkpoint index)."""
import pytest
import torch

import mos_path  # noqa: F401


def _count(sd, skip=()):
    return sum(v.numel() for k, v in sd.items() if not any(s in k for s in skip))


def test_unet_state_dict_matches_sd15_layout():
    from mixofshow.utils import pretrained
    sd = pretrained.load_unet('synthetic://sd15?seed=0').state_dict()
    assert _count(sd) == 859_520_964
    assert len(sd) == 686
    spot = {
        'conv_in.weight': (320, 4, 3, 3),
        'time_embedding.linear_1.weight': (1280, 320),
        'time_embedding.linear_2.bias': (1280, ),
        'down_blocks.0.resnets.0.norm1.weight': (320, ),
        'down_blocks.0.resnets.0.time_emb_proj.weight': (320, 1280),
        'down_blocks.0.attentions.0.proj_in.weight': (320, 320, 1, 1),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight': (320, 320),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias': (320, ),
        'down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight': (320, 768),
        'down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight': (2560, 320),
        'down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.weight': (320, 1280),
        'down_blocks.0.attentions.0.transformer_blocks.0.norm3.bias': (320, ),
        'down_blocks.0.downsamplers.0.conv.weight': (320, 320, 3, 3),
        'down_blocks.1.resnets.0.conv_shortcut.weight': (640, 320, 1, 1),
        'down_blocks.2.attentions.1.transformer_blocks.0.attn2.to_v.weight': (1280, 768),
        'down_blocks.3.resnets.1.conv2.weight': (1280, 1280, 3, 3),
        'mid_block.attentions.0.transformer_blocks.0.attn1.to_k.weight': (1280, 1280),
        'mid_block.resnets.1.norm2.bias': (1280, ),
        'up_blocks.0.resnets.0.conv1.weight': (1280, 2560, 3, 3),
        'up_blocks.0.upsamplers.0.conv.weight': (1280, 1280, 3, 3),
        'up_blocks.1.resnets.2.conv_shortcut.weight': (1280, 1920, 1, 1),
        'up_blocks.2.attentions.2.transformer_blocks.0.attn2.to_q.weight': (640, 640),
        'up_blocks.3.resnets.0.conv1.weight': (320, 960, 3, 3),
        'up_blocks.3.resnets.2.conv_shortcut.weight': (320, 640, 1, 1),
        'up_blocks.3.attentions.2.proj_out.bias': (320, ),
        'conv_norm_out.weight': (320, ),
        'conv_out.weight': (4, 320, 3, 3),
    }
    for k, shp in spot.items():
        assert k in sd, k
        assert tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape), shp)
    # no attention projection in the UNet carries a bias except to_out.0; every transformer block has exactly these leaves
    blk = sorted(k.split('transformer_blocks.0.')[1] for k in sd if k.startswith('mid_block.attentions.0.transformer_blocks.0.'))
    assert blk == sorted(['attn1.to_q.weight', 'attn1.to_k.weight', 'attn1.to_v.weight', 'attn1.to_out.0.weight',
                          'attn1.to_out.0.bias', 'attn2.to_q.weight', 'attn2.to_k.weight', 'attn2.to_v.weight',
                          'attn2.to_out.0.weight', 'attn2.to_out.0.bias', 'ff.net.0.proj.weight', 'ff.net.0.proj.bias',
                          'ff.net.2.weight', 'ff.net.2.bias', 'norm1.weight', 'norm1.bias', 'norm2.weight', 'norm2.bias',
                          'norm3.weight', 'norm3.bias'])


def test_vae_state_dict_matches_sd15_layout():
    from mixofshow.utils import pretrained
    sd = pretrained.load_vae('synthetic://sd15?seed=0').state_dict()
    assert _count(sd) == 83_653_863
    assert len(sd) == 248
    spot = {
        'encoder.conv_in.weight': (128, 3, 3, 3),
        'encoder.down_blocks.0.downsamplers.0.conv.weight': (128, 128, 3, 3),
        'encoder.down_blocks.1.resnets.0.conv_shortcut.weight': (256, 128, 1, 1),
        'encoder.mid_block.attentions.0.to_q.weight': (512, 512),
        'encoder.mid_block.attentions.0.to_out.0.bias': (512, ),
        'encoder.mid_block.attentions.0.group_norm.weight': (512, ),
        'encoder.conv_out.weight': (8, 512, 3, 3),
        'quant_conv.weight': (8, 8, 1, 1),
        'post_quant_conv.weight': (4, 4, 1, 1),
        'decoder.conv_in.weight': (512, 4, 3, 3),
        'decoder.up_blocks.0.upsamplers.0.conv.weight': (512, 512, 3, 3),
        'decoder.up_blocks.2.resnets.0.conv_shortcut.weight': (256, 512, 1, 1),
        'decoder.up_blocks.3.resnets.2.conv2.weight': (128, 128, 3, 3),
        'decoder.conv_out.weight': (3, 128, 3, 3),
    }
    for k, shp in spot.items():
        assert k in sd and tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape) if k in sd else None, shp)
    # diffusers < 0.18 spelling of the attention projections is remapped on load
    old = {k.replace('.to_q.', '.query.').replace('.to_k.', '.key.').replace('.to_v.', '.value.').replace(
        '.to_out.0.', '.proj_attn.'): (v[:, :, None, None] if ('attentions' in k and v.dim() == 2) else v) for k, v in sd.items()}
    assert set(pretrained.remap_vae_keys(old)) == set(sd)


def test_text_encoder_state_dict_matches_clip_vit_l14_layout():
    from mixofshow.utils import pretrained
    sd = pretrained.load_text_encoder('synthetic://sd15?seed=0').state_dict()
    assert _count(sd, skip=('position_ids', )) == 123_060_480
    assert len([k for k in sd if 'position_ids' not in k]) == 196
    spot = {
        'text_model.embeddings.token_embedding.weight': (49408, 768),
        'text_model.embeddings.position_embedding.weight': (77, 768),
        'text_model.encoder.layers.0.self_attn.q_proj.weight': (768, 768),
        'text_model.encoder.layers.0.self_attn.out_proj.bias': (768, ),
        'text_model.encoder.layers.11.mlp.fc1.weight': (3072, 768),
        'text_model.encoder.layers.11.mlp.fc2.weight': (768, 3072),
        'text_model.encoder.layers.5.layer_norm2.weight': (768, ),
        'text_model.final_layer_norm.bias': (768, ),
    }
    for k, shp in spot.items():
        assert k in sd and tuple(sd[k].shape) == shp, (k, shp)
    # transformers >= 5 drops the `text_model.` prefix: both spellings load
    new_style = {k[len('text_model.'):]: v for k, v in sd.items()}
    assert set(pretrained.remap_text_encoder_keys(new_style)) == {k for k in sd if 'position_ids' not in k}


def test_real_checkpoint_directory_roundtrip(tmp_path):
    """The diffusers directory layout (safetensors + config.json per sub-model) written by save_pretrained loads back
    through the same code path a real SD-1.5 download would take (pretrained.load_* on a directory)."""
    from mixofshow.pipelines.pipeline_edlora import StableDiffusionPipeline
    from mixofshow.utils import pretrained
    pipe = StableDiffusionPipeline.from_pretrained('synthetic://tiny?seed=3', torch_dtype=torch.float16)
    pipe.save_pretrained(str(tmp_path / 'm'))
    for name, loader in (('unet', pretrained.load_unet), ('vae', pretrained.load_vae), ('text_encoder', pretrained.load_text_encoder)):
        a, b = getattr(pipe, name).state_dict(), loader(str(tmp_path / 'm')).state_dict()
        assert set(a) == set(b)
        for k in a:
            assert torch.equal(a[k].float(), b[k].float()), (name, k)


def test_clip_text_tower_equals_transformers_clip_text_model():
    """The text tower is restated locally (mixofshow/models/clip.py keeps the module paths ED-LoRA checkpoints name);
    `transformers` IS installed here, so its semantics are pinned against the real thing: same random weights loaded into
    transformers.CLIPTextModel and into the local model -> identical hidden states (causal mask, quick-GELU MLP, pre-LN
    blocks, final LayerNorm, learned positions), fp32 on the CPU."""
    transformers = pytest.importorskip('transformers')
    from mixofshow.models.clip import CLIPTextModel
    cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3,
                                      num_attention_heads=4, max_position_embeddings=77, hidden_act='quick_gelu',
                                      eos_token_id=999, bos_token_id=998, pad_token_id=1)
    torch.manual_seed(0)
    theirs = transformers.CLIPTextModel(cfg).eval()
    ours = CLIPTextModel(vocab_size=1000, hidden_size=128, num_attention_heads=4, intermediate_size=512,
                         num_hidden_layers=3, max_position_embeddings=77).eval()
    sd = {(k if k.startswith('text_model.') else 'text_model.' + k): v for k, v in theirs.state_dict().items()}
    ours.load_state_dict(sd, strict=True)
    ids = torch.randint(2, 990, (3, 77), generator=torch.Generator().manual_seed(1))
    ids[:, 0] = 998
    ids[0, 10:] = 999
    ids[1, 40:] = 999
    ids[2, 76] = 999
    with torch.no_grad():
        ref = theirs(input_ids=ids).last_hidden_state
        got = ours(ids)[0]
    assert got.shape == ref.shape
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
