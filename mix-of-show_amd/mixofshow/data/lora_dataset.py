"""Training data for the ED-LoRA tune.

The reference's LoraDataset (mixofshow/data/lora_dataset.py:13-102 + pil_transform.py) is CPU data preparation built on
torchvision / cv2 (neither is installed here); it is a CALLER of the hot path. Two datasets are provided
behind the same `datasets.train` option block:
  * `SyntheticLoraDataset` (name: SyntheticLoraDataset, or concept_list: synthetic://...): seeded
    tensors of the shapes the trainer consumes — images U(-1,1) (3,512,512), masks = centred box of ones in
    (1,64,64), img_masks ones, captions with the concept tokens. Used by bench.py and the tests.
  * `LoraDataset`: the reference's dataset on a PIL + numpy restatement of its transforms (mixofshow.data.pil_transform:
    HumanResizeCropFinalV3, ResizeFillMaskNew, ShuffleCaption, EnhanceText, ToTensor, Normalize, ...), consuming Python's
    `random` and torch's RNG in the reference order.
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
    """Concept images + captions + masks for the ED-LoRA tune (reference mixofshow/data/lora_dataset.py:13-102): every
    file of each `instance_data_dir` (except .DS_Store), caption = first line of `<caption_dir>/<stem>.txt` when
    `use_caption`, mask = `<mask_dir>/<stem>.png` when `use_mask`; `replace_mapping` applied, runs of spaces squeezed; the
    list is shuffled once with Python's `random`; samples go through the `instance_transform` chain
    (mixofshow.data.pil_transform). Items: images (3,S,S), img_masks (1,S/8,S/8), prompts, and masks (1,S/8,S/8) only
    when a mask file is configured — without it the training loop falls back to img_masks, like the reference."""

    def __init__(self, opt):
        from mixofshow.data.pil_transform import PairCompose, build_transform
        self.opt = opt
        with open(opt['concept_list'], 'r') as f:
            concept_list = json.load(f)
        mapping = opt.get('replace_mapping', {}) or {}
        use_caption = opt.get('use_caption', False)
        use_mask = opt.get('use_mask', False)
        self.items = []
        for concept in concept_list:
            prompt = self.process_text(concept['instance_prompt'], mapping)
            caption_dir, mask_dir = concept.get('caption_dir'), concept.get('mask_dir')
            for fn in sorted(os.listdir(concept['instance_data_dir'])):       # sorted: a deterministic pre-shuffle order
                path = os.path.join(concept['instance_data_dir'], fn)
                if not os.path.isfile(path) or fn == '.DS_Store':
                    continue
                stem = os.path.splitext(fn)[0]
                cap = prompt
                if use_caption and caption_dir is not None:
                    cp = os.path.join(caption_dir, stem + '.txt')
                    if os.path.exists(cp):
                        with open(cp, 'r') as fr:
                            cap = self.process_text(fr.readlines()[0], mapping)
                mask = os.path.join(mask_dir, stem + '.png') if (use_mask and mask_dir is not None) else None
                self.items.append((path, cap, mask))
        random.shuffle(self.items)
        self.enlarge = int(opt.get('dataset_enlarge_ratio', 1))
        self.instance_transform = PairCompose([build_transform(t) for t in opt['instance_transform']])

    @staticmethod
    def process_text(text, mapping):
        for k, v in mapping.items():
            text = text.replace(k, v)
        return re.sub(' +', ' ', text.strip())

    def __len__(self):
        return len(self.items) * self.enlarge

    def __getitem__(self, index):
        from PIL import Image
        path, cap, mask = self.items[index % len(self.items)]
        extra = {'prompts': cap}
        if mask is not None:
            extra['mask'] = Image.open(mask).convert('L')
        img, extra = self.instance_transform(Image.open(path).convert('RGB'), **extra)
        if 'img_mask' not in extra:
            raise NotImplementedError('the instance_transform chain must produce `img_mask` '
                                      '(HumanResizeCropFinalV3 / ResizeFillMaskNew; reference lora_dataset.py:96-99)')
        out = {'images': img, 'img_masks': extra['img_mask'].unsqueeze(0).float(), 'prompts': extra['prompts']}
        if 'mask' in extra:
            out['masks'] = extra['mask'].unsqueeze(0).float()
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
