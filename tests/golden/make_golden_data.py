"""Golden vectors of the DATA transforms, produced by EXECUTING the reference's own classes
(/root/reference/mixofshow/data/pil_transform.py and lora_dataset.py, read-only, unmodified):

  G9  ShuffleCaption / EnhanceText (:257-364)     pure Python -- nothing but the reference runs
  G10 HumanResizeCropFinalV3 (:125-195), ResizeFillMaskNew (:198-254), PairRandomCrop, PairCompose
  G11 LoraDataset.__getitem__ through the shipped `instance_transform` chain of options/train/EDLoRA/real/8101_*.yml

torchvision and cv2 are not installable here; for G10 / G11 the handful of their functions these classes call comes from
oracle/vision_shim.py (PIL's own `Image.resize` / `crop` behind torchvision's size rule; cv2's default bilinear resize), the
same arrangement as oracle/attention_shim.py for diffusers. What the goldens therefore pin is everything the REFERENCE wrote:
the order in which Python's `random` and torch's generator are consumed, the crop-branch logic, canvas placement, the
`/255` and 1/8-mask steps, caption handling, dataset item layout.

Runs only where /root/reference exists.  Usage:  python tests/golden/make_golden_data.py [output.pt]
"""
import hashlib
import json
import os
import random
import sys
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.normpath(os.path.join(HERE, '..', '..'))
REF = '/root/reference'


def _install_vision_modules():
    sys.path.insert(0, REPO)
    from oracle import vision_shim as V
    tv = types.ModuleType('torchvision')
    tr = types.ModuleType('torchvision.transforms')
    fn = types.ModuleType('torchvision.transforms.functional')
    for name in ('resize', 'crop', 'hflip', 'to_tensor', 'InterpolationMode'):
        setattr(fn, name, getattr(V, name))
    for name in ('CenterCrop', 'Normalize', 'RandomCrop', 'RandomHorizontalFlip', 'Resize', 'InterpolationMode'):
        setattr(tr, name, getattr(V, name))
    tr.functional = fn
    tv.transforms = tr
    cv2 = types.ModuleType('cv2')
    cv2.resize, cv2.INTER_NEAREST, cv2.INTER_LINEAR = V.cv2_resize, V.INTER_NEAREST, V.INTER_LINEAR
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tr, 'torchvision.transforms.functional': fn, 'cv2': cv2})


def import_reference_data():
    _install_vision_modules()
    sys.path.insert(0, REF)
    from mixofshow.data import lora_dataset as ref_ds
    from mixofshow.data import pil_transform as ref_t
    assert ref_t.__file__.startswith(REF) and ref_ds.__file__.startswith(REF), (ref_t.__file__, ref_ds.__file__)
    return ref_t, ref_ds


# ---- seeded inputs (shared with tests/test_data_golden.py) -------------------------------------------------------------
def photo(w, h, seed):
    """A smooth seeded RGB image with structure at several scales (so that resize / crop offsets show up in every pixel)."""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(4):
            fx, fy, ph = rs.uniform(0.005, 0.08), rs.uniform(0.005, 0.08), rs.uniform(0, 6.28)
            img[:, :, c] += np.sin(fx * x + fy * y + ph)
    img = (img - img.min()) / (img.max() - img.min())
    return Image.fromarray((img * 255).astype(np.uint8))


def person_mask(w, h, seed):
    rs = np.random.RandomState(1000 + seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = rs.uniform(0.35, 0.65) * w, rs.uniform(0.35, 0.65) * h
    rx, ry = rs.uniform(0.15, 0.3) * w, rs.uniform(0.2, 0.4) * h
    return Image.fromarray((((x - cx) / rx)**2 + ((y - cy) / ry)**2 <= 1).astype(np.uint8) * 255)


CAPTIONS = ['<TOK>, a man in a black suit, standing, looking at viewer , outdoors', 'a photo of <TOK>',
            '  <TOK>,red scarf,  glasses, castle in the background,night  ', '<TOK>']
GEOMETRY_CASES = [(768, 768), (600, 900), (900, 600), (512, 512), (1024, 520), (515, 1400)]     # (w, h)


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes()).hexdigest()


def thumb(img):
    """1/8 box means of an RGB uint8 image: a readable stand-in next to the hash."""
    a = np.asarray(img, dtype=np.float64)
    h, w = a.shape[0] // 8 * 8, a.shape[1] // 8 * 8
    return torch.from_numpy(a[:h, :w].reshape(h // 8, 8, w // 8, 8, 3).mean((1, 3)).round().astype(np.uint8))


def seed_all(s):
    random.seed(s)
    torch.manual_seed(s)


def caption_cases(T):
    out = []
    for keep in (0, 1, 2):
        for ci, cap in enumerate(CAPTIONS):
            for s in range(3):
                seed_all(100 * keep + 10 * ci + s)
                _, kw = T.ShuffleCaption(keep_token_num=keep).forward(None, prompts=cap)
                out.append(dict(kind='shuffle', keep=keep, caption=cap, seed=100 * keep + 10 * ci + s, out=kw['prompts']))
    for et in ('object', 'style', 'human'):
        for s in range(12):
            seed_all(s)
            _, kw = T.EnhanceText(enhance_type=et).forward(None, prompts='  <potter1> <potter2> ')
            out.append(dict(kind='enhance', enhance_type=et, seed=s, out=kw['prompts']))
    return out


def geometry_cases(T):
    out = []
    for gi, (w, h) in enumerate(GEOMETRY_CASES):
        for with_mask in (True, False):
            for s in range(3):
                for cls, kw in (('HumanResizeCropFinalV3', dict(size=512, crop_p=0.5)),
                                ('ResizeFillMaskNew', dict(size=512, crop_p=0.5, scale_ratio=[0.75, 1.0]))):
                    seed = 1000 * gi + 10 * s + (1 if with_mask else 0)
                    seed_all(seed)
                    extra = {'mask': person_mask(w, h, gi)} if with_mask else {}
                    img, res = getattr(T, cls)(**kw)(photo(w, h, gi), **extra)
                    # bit-exactness is carried by the hashes; the half-precision previews (first seed only) are there to
                    # make a mismatch readable
                    rec = dict(cls=cls, kwargs=kw, w=w, h=h, image_seed=gi, with_mask=with_mask, seed=seed, size=img.size,
                               image_sha=sha(np.asarray(img)), img_mask_sha=sha(res['img_mask'].numpy()),
                               img_mask_sum=float(res['img_mask'].sum()), rng_after=(random.random(), float(torch.rand(1))))
                    if with_mask:
                        rec.update(mask_sha=sha(res['mask'].numpy()), mask_sum=float(res['mask'].sum()))
                    if s == 0 and with_mask:
                        rec.update(image_thumb=thumb(img), img_mask_preview=res['img_mask'].half(), mask_preview=res['mask'].half())
                    out.append(rec)
    return out


def write_concept_folder(root):
    """Three concepts, ONE image each (the reference lists a folder with Path.iterdir(), whose order is the file system's)."""
    cfg = []
    for i, (w, h) in enumerate(((640, 960), (960, 640), (700, 700))):
        d = os.path.join(root, f'c{i}')
        os.makedirs(os.path.join(d, 'image'))
        os.makedirs(os.path.join(d, 'caption'))
        os.makedirs(os.path.join(d, 'mask'))
        photo(w, h, 20 + i).save(os.path.join(d, 'image', 'a.png'))
        person_mask(w, h, 20 + i).save(os.path.join(d, 'mask', 'a.png'))
        with open(os.path.join(d, 'caption', 'a.txt'), 'w') as f:
            f.write(CAPTIONS[i] + '\nsecond line is ignored\n')
        cfg.append(dict(instance_prompt='<TOK>', instance_data_dir=os.path.join(d, 'image'),
                        caption_dir=os.path.join(d, 'caption'), mask_dir=os.path.join(d, 'mask')))
    path = os.path.join(root, 'concepts.json')
    with open(path, 'w') as f:
        json.dump(cfg, f)
    return path


def dataset_opt(concept_list, use_mask=True):
    # options/train/EDLoRA/real/8101_EDLoRA_potter_Cmix_B4_Repeat500.yml:9-22
    return dict(name='LoraDataset', concept_list=concept_list, use_caption=True, use_mask=use_mask,
                instance_transform=[dict(type='HumanResizeCropFinalV3', size=512, crop_p=0.5), dict(type='ToTensor'),
                                    dict(type='Normalize', mean=[0.5], std=[0.5]),
                                    dict(type='ShuffleCaption', keep_token_num=1), dict(type='EnhanceText', enhance_type='human')],
                replace_mapping={'<TOK>': '<potter1> <potter2>'}, batch_size_per_gpu=2, dataset_enlarge_ratio=4)


def dataset_cases(DS, root):
    out = []
    for use_mask in (True, False):
        seed_all(7)
        ds = DS.LoraDataset(dataset_opt(write_concept_folder(os.path.join(root, f'm{int(use_mask)}')), use_mask))
        items = []
        for i in range(len(ds)):
            ex = ds[i]
            items.append({k: (v if isinstance(v, str) else dict(sha=sha(v.numpy()), dtype=str(v.dtype), shape=tuple(v.shape),
                                                                mean=float(v.double().mean())))
                          for k, v in ex.items()})
        out.append(dict(use_mask=use_mask, length=len(ds), items=items))
    return out


def main(out_path=None):
    import tempfile
    T, DS = import_reference_data()
    with tempfile.TemporaryDirectory() as root:
        out = dict(captions=caption_cases(T), geometry=geometry_cases(T), dataset=dataset_cases(DS, root),
                   source='reference mixofshow/data/pil_transform.py + lora_dataset.py executed; torchvision / cv2 functions from '
                          'oracle/vision_shim.py')
    path = out_path or os.path.join(HERE, 'reference_data_golden.pt')
    torch.save(out, path)
    print(f'wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB): {len(out["captions"])} caption, {len(out["geometry"])} '
          f'geometry, {sum(len(d["items"]) for d in out["dataset"])} dataset cases')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else None)
