"""train_edlora.py / test_edlora.py at CLI level on the CPU ('tiny' preset, HIP primitives emulated): option file ->
datasets -> TrainEngine loop -> checkpoint -> validation sampling with the merged LoRA (SURVEY rows A7/A8 host side)."""
import argparse
import os

import torch
import yaml


def _recipe(tmp_path, total_images=4, val=True):
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'options', 'train', 'EDLoRA',
                           'synthetic', '8101_EDLoRA_potter_synthetic_B4.yml')) as f:
        opt = yaml.safe_load(f)
    opt['name'] = 'cli_cpu'
    opt['mixed_precision'] = 'no'
    opt['models']['pretrained_path'] = 'synthetic://tiny?seed=0'
    tr = opt['datasets']['train']
    tr.update(num_images=total_images, dataset_enlarge_ratio=2, batch_size_per_gpu=2)
    tr['instance_transform'][0]['size'] = 64
    opt['datasets']['val_vis'].update(latent_size=[4, 8, 8], num_samples_per_prompt=1, batch_size_per_gpu=1)
    opt['val'].update(val_during_save=val, alpha_list=[0.7], sample=dict(num_inference_steps=2, guidance_scale=7.5))
    opt['logger'] = dict(print_freq=1, save_checkpoint_freq=1000)
    p = tmp_path / 'recipe.yml'
    with open(p, 'w') as f:
        yaml.safe_dump(opt, f)
    return str(p)


def test_train_cli_then_validation(emulated_hip, tmp_path):
    import train_edlora
    recipe = _recipe(tmp_path)
    train_edlora.train(str(tmp_path), argparse.Namespace(opt=recipe))
    ckpt = tmp_path / 'experiments' / 'cli_cpu' / 'models' / 'edlora_model-latest.pth'
    assert ckpt.exists()
    sd = torch.load(ckpt, weights_only=False)['params']
    assert set(sd) == {'new_concept_embedding', 'text_encoder', 'unet'}
    assert sd['new_concept_embedding']['<potter1>'].shape == (16, 64)
    # 4 images x enlarge 2 / batch 2 = 4 optimisation steps: the LoRA `up` factors have left their zero init
    ups = [v for k, v in sd['unet'].items() if k.endswith('lora_up.weight')]
    assert ups and all(torch.isfinite(u).all() for u in ups) and any(u.abs().max() > 0 for u in ups)
    # validation images written by the save hook
    vis = tmp_path / 'experiments' / 'cli_cpu' / 'visualization'
    pngs = [os.path.join(d, f) for d, _, fs in os.walk(vis) for f in fs if f.endswith(('.png', '.jpg'))]
    assert pngs, 'no validation image written'


def test_test_cli_on_saved_checkpoint(emulated_hip, tmp_path):
    """test_edlora.py: load the ED-LoRA checkpoint, merge it at alpha, sample the validation prompts, write PNGs."""
    import test_edlora
    import train_edlora
    recipe = _recipe(tmp_path, val=False)
    train_edlora.train(str(tmp_path), argparse.Namespace(opt=recipe))
    with open(recipe) as f:
        opt = yaml.safe_load(f)
    opt['name'] = 'cli_cpu_test'
    opt['path'] = dict(lora_path=str(tmp_path / 'experiments' / 'cli_cpu' / 'models' / 'edlora_model-latest.pth'))
    opt['models']['alpha'] = 0.7
    p = tmp_path / 'test.yml'
    with open(p, 'w') as f:
        yaml.safe_dump(opt, f)
    opt['val']['compose_visualize'] = True
    with open(p, 'w') as f:
        yaml.safe_dump(opt, f)
    test_edlora.test(str(tmp_path), argparse.Namespace(opt=str(p)))
    # the reference's layout (test_edlora.py:44): <visualization>/<dataset name>/<current_iter>/<prompt>---G_x_S_y---<i>---<iter>.png
    out = tmp_path / 'results' / 'cli_cpu_test' / 'visualization' / opt['datasets']['val_vis']['name'] / 'validation_0.7'
    pngs = [f for f in os.listdir(out) if f.endswith('.png')]
    assert len(pngs) == 1 and '<potter1>' in pngs[0]       # 1 prompt x 1 sample, <TOK> replaced in the file name
    assert pngs[0].endswith('---G_7.5_S_2---1---validation_0.7.png')
    # compose_visualize (reference utils/util.py:279-313): caption tile + the sample, one row per prompt, saved beside the folder
    grid = out.parent / 'G_7.5_S_2---validation_0.7.jpg'
    assert grid.exists()
    from PIL import Image
    w, h = Image.open(grid).size
    assert (w, h) == (2 * (64 + 2) + 2, 64 + 2 + 2)        # make_grid: 2 tiles of 64 px (8x8 latent), 2-pixel padding


def test_compose_visualize_grid_layout(tmp_path):
    """Two prompts x two samples -> 2 rows of [caption | sample | sample]; mixed sample args are refused (reference assert)."""
    import numpy as np
    import pytest
    from PIL import Image
    import mos_path  # noqa: F401
    from mixofshow.utils.util import compose_visualize, make_grid
    d = tmp_path / 'val' / 'iter1'
    d.mkdir(parents=True)
    for pi, prompt in enumerate(('a_cat', 'a_dog')):
        for i in (1, 2):
            Image.fromarray(np.full((32, 48, 3), 40 * (2 * pi + i), dtype=np.uint8)).save(d / f'{prompt}---G_7.5_S_50---{i}---iter1.png')
    out = compose_visualize(str(d))
    assert out.endswith('G_7.5_S_50---iter1.jpg')
    img = np.asarray(Image.open(out).convert('RGB')).astype(int)
    assert img.shape == (2 * 34 + 2, 3 * 50 + 2, 3)
    assert abs(img[2 + 16, 2 + 50 + 24].mean() - 40) <= 3 and abs(img[2 + 34 + 16, 2 + 100 + 24].mean() - 160) <= 3
    assert img[0].max() <= 8                                # padding rows are black
    g = make_grid([torch.ones(3, 4, 4)] * 5, nrow=2)
    assert g.shape == (3, 3 * 6 + 2, 2 * 6 + 2)
    Image.fromarray(np.zeros((32, 48, 3), dtype=np.uint8)).save(d / 'a_cat---G_3_S_50---3---iter1.png')
    with pytest.raises(AssertionError, match='same sample args'):
        compose_visualize(str(d))
