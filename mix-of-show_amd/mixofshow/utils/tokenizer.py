"""Tokenizer access. With real SD-1.5 files on disk (`<pretrained>/tokenizer/{vocab.json,merges.txt}`) the
transformers CLIPTokenizer is used. The authoring/benchmark environment has no tokenizer files and no
network, so `SyntheticCLIPTokenizer` provides the same call surface (padding to 77 with EOS, BOS/EOS ids,
`add_tokens`, `convert_tokens_to_ids`, `encode`) over a deterministic word->id hash; concept tokens added with
`add_tokens` get ids from 49408 upwards exactly like the real tokenizer (reference trainer_edlora.py:160-164)."""
import os
import re
import zlib
from types import SimpleNamespace

import torch

BOS, EOS = 49406, 49407


class SyntheticCLIPTokenizer:
    model_max_length = 77

    def __init__(self):
        self.added = {}
        self._base = 49408

    def __len__(self):
        return self._base + len(self.added)

    def add_tokens(self, new_tokens):
        if isinstance(new_tokens, str):
            new_tokens = [new_tokens]
        n = 0
        for t in new_tokens:
            if t not in self.added:
                self.added[t] = self._base + len(self.added)
                n += 1
        return n

    def convert_tokens_to_ids(self, token):
        if isinstance(token, (list, tuple)):
            return [self.convert_tokens_to_ids(t) for t in token]
        if token in self.added:
            return self.added[token]
        return 1000 + zlib.crc32(token.lower().encode('utf-8')) % 48000

    def _split(self, text):
        if self.added:
            pat = '(' + '|'.join(re.escape(t) for t in sorted(self.added, key=len, reverse=True)) + ')'
            parts = re.split(pat, text)
        else:
            parts = [text]
        toks = []
        for p in parts:
            if p in self.added:
                toks.append(p)
            else:
                toks += re.findall(r"[A-Za-z0-9]+|[^\sA-Za-z0-9]", p)
        return toks

    def encode(self, text, add_special_tokens=True):
        ids = [self.convert_tokens_to_ids(t) for t in self._split(text)]
        return [BOS] + ids + [EOS] if add_special_tokens else ids

    def __call__(self, text, padding='max_length', max_length=None, truncation=True, return_tensors=None, **kw):
        single = isinstance(text, str)
        texts = [text] if single else list(text)
        L = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = self.encode(t)
            if truncation and len(ids) > L:
                ids = ids[:L - 1] + [EOS]
            if padding == 'max_length':
                ids = ids + [EOS] * (L - len(ids))
            rows.append(ids)
        if return_tensors == 'pt':
            return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))
        return SimpleNamespace(input_ids=rows[0] if single else rows)


def load_tokenizer(pretrained_path):
    tok_dir = os.path.join(str(pretrained_path), 'tokenizer')
    if os.path.isfile(os.path.join(tok_dir, 'vocab.json')):
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(tok_dir)
    tok = SyntheticCLIPTokenizer()
    added = os.path.join(tok_dir, 'added_tokens.json')      # written by StableDiffusionPipeline.save_pretrained
    if os.path.isfile(added):
        import json
        with open(added) as f:
            tok.added = {t: int(i) for t, i in sorted(json.load(f).items(), key=lambda kv: kv[1])}
    return tok
