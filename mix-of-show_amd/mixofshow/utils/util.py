"""Validation-image helpers of the reference's mixofshow/utils/util.py that its entry points import
(test_edlora.py:18: NEGATIVE_PROMPT, compose_visualize, pil_imwrite): file naming and the comparison grid. PIL + torch only
(the reference builds the grid with torchvision's ToTensor / make_grid and draws captions with a bundled arial.ttf; neither is a
dependency here: `make_grid` is restated -- 2-pixel black padding, `nrow` images per row -- and the caption uses PIL's default
font when no TrueType file is given)."""
import os
import os.path as osp

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageFont

# reference mixofshow/utils/util.py:17
NEGATIVE_PROMPT = ('longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality, '
                   'low quality')


def pil_imwrite(img, file_path, auto_mkdir=True):
    """reference :232-248"""
    assert isinstance(img, Image.Image), 'model should return a list of PIL images'
    if auto_mkdir:
        os.makedirs(osp.abspath(osp.dirname(file_path)), exist_ok=True)
    img.save(file_path)


def draw_prompt(text, height, width, font_size=45, font_path=None):
    """White tile with the prompt wrapped to 80 % of the width, starting at (10 %, 30 %) (reference :251-276)."""
    img = Image.new('RGB', (width, height), (255, 255, 255))
    draw = ImageDraw.Draw(img)
    try:
        font = ImageFont.truetype(font_path, font_size) if font_path else ImageFont.load_default(font_size)
    except (OSError, TypeError):
        font = ImageFont.load_default()
    per_line = 0
    while per_line < len(text) and draw.textlength(text[:per_line], font=font) + 0.1 * width < width - 0.1 * width:
        per_line += 1
    per_line = max(per_line, 1)
    out = ''
    for idx, ch in enumerate(text):
        if idx % per_line == 0:
            out += '\n'
            if ch == ' ':
                ch = ''
        out += ch
    draw.text([int(0.1 * width), int(0.3 * height)], out, font=font, fill='black')
    return img


def _to_tensor(img):
    a = np.asarray(img.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(a.transpose(2, 0, 1).copy()).float().div(255)


def make_grid(tensors, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid for a list of equally sized (C, H, W) tensors."""
    t = torch.stack(list(tensors))
    n, c, h, w = t.shape
    xmaps = min(nrow, n)
    ymaps = -(-n // xmaps)
    hh, ww = h + padding, w + padding
    grid = t.new_full((c, hh * ymaps + padding, ww * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= n:
                break
            grid[:, y * hh + padding:y * hh + padding + h, x * ww + padding:x * ww + padding + w] = t[k]
            k += 1
    return grid


def compose_visualize(dir_path):
    """reference :279-313 — one row per prompt: a caption tile followed by that prompt's samples; the files of `dir_path` are named
    `{prompt}---{sample_args}---{index}---{suffix}.png`; the grid is written next to the directory as
    `{sample_args}---{suffix}.jpg`. Returns its path."""
    img_list, prompts, sample_args, suffixes = [], [], set(), set()
    for filename in sorted(os.listdir(dir_path)):
        prompt, args, _index, suffix = osp.splitext(osp.basename(filename))[0].split('---')
        img = _to_tensor(Image.open(osp.join(dir_path, filename)))
        height, width = img.shape[1:]
        if prompt not in prompts:
            img_list.append(_to_tensor(draw_prompt(prompt, height=height, width=width, font_size=45)))
            prompts.append(prompt)
        sample_args.add(args)
        suffixes.add(suffix)
        img_list.append(img)
    assert len(sample_args) == 1, 'compose dir should contain images form same sample args.'
    assert len(suffixes) == 1, 'compose dir should contain images form same suffix.'
    grid = make_grid(img_list, nrow=len(img_list) // len(prompts))
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to('cpu', torch.uint8).numpy()
    out = osp.join(osp.dirname(dir_path), f'{sample_args.pop()}---{suffixes.pop()}.jpg')
    Image.fromarray(arr).save(out)
    return out
