"""Training data for the ED-LoRA tune.

The reference's LoraDataset (mixofshow/data/lora_dataset.py:13-102 + pil_transform.py) is CPU data preparation
built on torchvision / cv2 (neither is installed here) and is OUT OF SCOPE of the hot path. Two
datasets are provided behind the same `datasets.train` option block:
  * `SyntheticLoraDataset` (name: SyntheticLoraDataset, or concept_list: synthetic://...): seeded
    tensors of the shapes the trainer consumes — images U(-1,1) (3,512,512), masks = centred box of ones in
    (1,64,64), img_masks ones, captions with the concept tokens. Used by bench.py and the tests.
  * `LoraDataset`: a PIL-only loader for real concept folders supporting resize + centre-crop to `size`, ToTensor,
    Normalize, caption files, mask folders and `replace_mapping`. The reference's augmentation transforms
    (HumanResizeCropFinalV3, ShuffleCaption, EnhanceText, ...) are accepted in the option list and mapped to the
    deterministic centre-crop pipeline; their random behaviour is not reproduced.
"""
import json
import os
import random
import re

import torch
from torch.utils.data import Dataset


class SyntheticLoraDataset(Dataset):

    def __init__(self, opt):
        self.opt = opt
        self.size = int(opt.get('image_size', 512))
        self.length = int(opt.get('num_images', 8)) * int(opt.get('dataset_enlarge_ratio', 1))
        mapping = opt.get('replace_mapping', {'<TOK>': '<potter1> <potter2>'}) or {}
        self.caption = 'a <TOK> in the park, 4K, high quality'
        for k, v in mapping.items():
            self.caption = self.caption.replace(k, v)
        self.seed = int(opt.get('seed', 0))

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 100003 + index)
        s, m = self.size, self.size // 8
        images = torch.rand(3, s, s, generator=g) * 2 - 1
        masks = torch.zeros(1, m, m)
        masks[:, m // 4:3 * m // 4, m // 4:3 * m // 4] = 1
        return {'images': images, 'masks': masks, 'img_masks': torch.ones(1, m, m), 'prompts': self.caption}


class LoraDataset(Dataset):

    def __init__(self, opt):
        from PIL import Image  # noqa: F401
        self.opt = opt
        with open(opt['concept_list'], 'r') as f:
            concept_list = json.load(f)
        self.size = 512
        for t in opt.get('instance_transform', []):
            if 'size' in t:
                self.size = int(t['size'])
        self.use_caption = opt.get('use_caption', False)
        self.use_mask = opt.get('use_mask', False)
        self.mapping = opt.get('replace_mapping', {}) or {}
        self.items = []
        for concept in concept_list:
            d = concept['instance_data_dir']
            for fn in sorted(os.listdir(d)):
                if not fn.lower().endswith(('.png', '.jpg', '.jpeg', '.webp')):
                    continue
                stem = os.path.splitext(fn)[0]
                cap = concept['instance_prompt']
                if self.use_caption and concept.get('caption_dir'):
                    cp = os.path.join(concept['caption_dir'], stem + '.txt')
                    if os.path.exists(cp):
                        cap = open(cp).read().strip()
                mask = None
                if self.use_mask and concept.get('mask_dir'):
                    mp = os.path.join(concept['mask_dir'], stem + '.png')
                    mask = mp if os.path.exists(mp) else None
                self.items.append((os.path.join(d, fn), cap, mask))
        random.shuffle(self.items)
        self.enlarge = int(opt.get('dataset_enlarge_ratio', 1))

    def __len__(self):
        return len(self.items) * self.enlarge

    def _load(self, path, mode, size, nearest=False):
        import numpy as np
        from PIL import Image
        im = Image.open(path).convert(mode)
        w, h = im.size
        s = size / min(w, h)
        im = im.resize((max(size, round(w * s)), max(size, round(h * s))), Image.NEAREST if nearest else Image.BICUBIC)
        w, h = im.size
        l, t = (w - size) // 2, (h - size) // 2
        im = im.crop((l, t, l + size, t + size))
        return torch.from_numpy(np.array(im)).float() / 255.0

    def __getitem__(self, index):
        path, cap, mask = self.items[index % len(self.items)]
        img = self._load(path, 'RGB', self.size).permute(2, 0, 1) * 2 - 1
        for k, v in self.mapping.items():
            cap = cap.replace(k, v)
        cap = re.sub(' +', ' ', cap.strip())
        m = self.size // 8
        out = {'images': img, 'prompts': cap, 'img_masks': torch.ones(1, m, m)}
        if mask is not None:
            mk = self._load(mask, 'L', self.size, nearest=True)[None]
            out['masks'] = torch.nn.functional.interpolate(mk[None], size=(m, m), mode='nearest')[0]
        # no mask file: the key is omitted (reference lora_dataset.py:90-94) and the loop falls back to img_masks
        return out


def build_train_dataset(opt):
    name = opt.get('name', 'LoraDataset')
    concept_list = str(opt.get('concept_list', ''))
    if name == 'SyntheticLoraDataset' or concept_list.startswith('synthetic://'):
        return SyntheticLoraDataset(opt)
    if not os.path.exists(concept_list):
        # a typo must not silently train on noise
        raise FileNotFoundError(f'datasets.train.concept_list {concept_list!r} does not exist (use name: '
                                'SyntheticLoraDataset or concept_list: synthetic://... for the seeded synthetic data)')
    return LoraDataset(opt)
