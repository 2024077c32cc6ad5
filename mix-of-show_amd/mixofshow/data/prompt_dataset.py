"""PromptDataset — validation prompts x N samples with deterministic CPU latents
`torch.randn(latent_size, generator=torch.manual_seed(idx))` (reference mixofshow/data/prompt_dataset.py:9-67).
The latent recipe is platform independent and is the parity fixture for sampling."""
import os
import random
import re

import torch
from torch.utils.data import Dataset


class PromptDataset(Dataset):

    def __init__(self, opt):
        self.opt = opt
        prompts = opt['prompts']
        if isinstance(prompts, str):
            if not os.path.exists(prompts):
                raise ValueError('prompts should be a prompt file path or prompt list, please check!')
            with open(prompts, 'r') as f:
                prompts = [l.strip() for l in f.readlines()]
        mapping = opt.get('replace_mapping', {}) or {}
        cleaned = []
        for line in prompts:
            if len(line.strip()) == 0:
                continue
            for k, v in mapping.items():
                line = line.replace(k, v)
            cleaned.append(re.sub(' +', ' ', line.strip()))
        self.prompts = cleaned
        self.num_samples_per_prompt = opt['num_samples_per_prompt']
        self.prompts_to_generate = [(p, i) for i in range(1, self.num_samples_per_prompt + 1) for p in self.prompts]
        self.latent_size = tuple(opt['latent_size'])
        self.share_latent_across_prompt = opt.get('share_latent_across_prompt', True)

    def __len__(self):
        return len(self.prompts_to_generate)

    def __getitem__(self, index):
        prompt, indice = self.prompts_to_generate[index]
        seed = indice if self.share_latent_across_prompt else random.randint(0, 1000)
        return {'prompts': prompt, 'indices': indice,
                'latents': torch.randn(self.latent_size, generator=torch.manual_seed(seed))}
