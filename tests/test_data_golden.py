"""SURVEY 8(f).4 pinned to the reference (VERDICT r04 missing #4): the product's data transforms and LoraDataset against
goldens made by EXECUTING the reference's own classes (tests/golden/make_golden_data.py -> reference_data_golden.pt).
Byte / index work: bit-exact (hashes of the uint8 image, of the float64 1/8 masks, the caption strings, and the state of both
random generators afterwards -- i.e. the same number of draws in the same order)."""
import importlib.util
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
import mos_path  # noqa: E402,F401
from mixofshow.data import lora_dataset as DS  # noqa: E402
from mixofshow.data import pil_transform as T  # noqa: E402


def _recipe():
    spec = importlib.util.spec_from_file_location('make_golden_data', os.path.join(GOLDEN_DIR, 'make_golden_data.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope='module')
def gold():
    return torch.load(os.path.join(GOLDEN_DIR, 'reference_data_golden.pt'), weights_only=False)


def test_caption_transforms_vs_reference_golden(gold):
    R = _recipe()
    for c in gold['captions']:
        R.seed_all(c['seed'])
        if c['kind'] == 'shuffle':
            _, kw = T.ShuffleCaption(keep_token_num=c['keep']).forward(None, prompts=c['caption'])
        else:
            _, kw = T.EnhanceText(enhance_type=c['enhance_type']).forward(None, prompts='  <potter1> <potter2> ')
        assert kw['prompts'] == c['out'], c
    with pytest.raises(NotImplementedError):
        T.EnhanceText(enhance_type='texture')            # reference :353-354


def test_geometric_transforms_vs_reference_golden(gold):
    R = _recipe()
    branches = set()
    for c in gold['geometry']:
        R.seed_all(c['seed'])
        extra = {'mask': R.person_mask(c['w'], c['h'], c['image_seed'])} if c['with_mask'] else {}
        img, res = getattr(T, c['cls'])(**c['kwargs']).forward(R.photo(c['w'], c['h'], c['image_seed']), **extra)
        tag = f"{c['cls']} {c['w']}x{c['h']} mask={c['with_mask']} seed={c['seed']}"
        assert img.size == tuple(c['size']), tag
        assert res['img_mask'].dtype == torch.float64 and float(res['img_mask'].sum()) == c['img_mask_sum'], tag
        assert R.sha(res['img_mask'].numpy()) == c['img_mask_sha'], tag
        if 'mask_preview' in c:
            assert torch.equal(res['mask'].half(), c['mask_preview']), tag
            assert torch.equal(R.thumb(img), c['image_thumb']), tag
        if c['with_mask']:
            assert R.sha(res['mask'].numpy()) == c['mask_sha'], tag
        else:
            assert 'mask' not in res
        assert R.sha(np.asarray(img)) == c['image_sha'], tag
        # both generators consumed exactly as the reference consumes them
        assert (random.random(), float(torch.rand(1))) == tuple(c['rng_after']), tag
        branches.add((c['cls'], c['w'] < c['h'], c['with_mask'], c['img_mask_sum'] == 4096.0))
    assert len(branches) >= 10           # portrait / landscape x mask / no mask x full / partial canvas, both classes


def test_lora_dataset_items_vs_reference_golden(gold, tmp_path):
    R = _recipe()
    for case in gold['dataset']:
        R.seed_all(7)
        ds = DS.LoraDataset(R.dataset_opt(R.write_concept_folder(str(tmp_path / f"m{int(case['use_mask'])}")), case['use_mask']))
        assert len(ds) == case['length']
        for i, want in enumerate(case['items']):
            ex = ds[i]
            assert set(ex) == set(want), (i, set(ex), set(want))
            for k, w in want.items():
                if isinstance(w, str):
                    assert ex[k] == w, (i, k)
                else:
                    # the reference hands the 1/8 masks on as float64 (lora_dataset.py:88-97); the product casts them to
                    # float32 for the trainer -- compared on the float64 values they came from
                    v = ex[k]
                    assert tuple(v.shape) == tuple(w['shape']), (i, k)
                    if k == 'images':
                        assert str(v.dtype) == w['dtype'] and R.sha(v.numpy()) == w['sha'], (i, k)
                    else:
                        assert abs(float(v.double().mean()) - w['mean']) < 1e-7, (i, k)


@pytest.mark.skipif(not os.path.isdir('/root/reference/mixofshow'), reason='reference tree not present')
def test_reference_data_golden_regenerates_bit_identical(tmp_path):
    out = tmp_path / 'd.pt'
    r = subprocess.run([sys.executable, os.path.join(GOLDEN_DIR, 'make_golden_data.py'), str(out)], cwd='/tmp',
                       env=dict(os.environ, PYTHONHASHSEED='0'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    from tests.test_reference_golden_recipe import _diff
    new, old = torch.load(out, weights_only=False), torch.load(os.path.join(GOLDEN_DIR, 'reference_data_golden.pt'), weights_only=False)
    assert _diff(new, old) == []
