"""TEST INFRASTRUCTURE ONLY (never imported by the product): the handful of torchvision / cv2 functions the reference's data
transforms call (mixofshow/data/pil_transform.py:5-12), restated for PIL inputs so that tests/golden/make_golden_data.py can
EXECUTE the reference's own transform classes (neither package is installable here).

Restated from the published behaviour of torchvision 0.15 (`transforms/functional.py`, `_functional_pil.py`,
`transforms.RandomCrop.get_params`) and OpenCV 4 (`resize`, INTER_LINEAR on CV_64F) -- "from memory", like
oracle/attention_shim.py for diffusers:

  * F.resize on a PIL image = `Image.resize` with the output size of `_compute_resized_output_size`: an int `size` is the
    SHORTER edge, the longer one is int(size * long / short); with `max_size`, when the longer edge would exceed it the pair
    becomes (int(max_size * short' / long'), max_size). A (h, w) pair is taken as is. Interpolation default BILINEAR; the int
    0 the reference passes for masks (:219) is the legacy spelling of NEAREST.
  * F.crop(img, top, left, height, width) = `img.crop((left, top, left + width, top + height))`.
  * RandomCrop draws `i = torch.randint(0, h - th + 1, (1,))` then `j = torch.randint(0, w - tw + 1, (1,))` from torch's
    global generator, and nothing when the sizes already match.
  * cv2.resize(src, (w, h), <third positional argument>): the third positional parameter is `dst`, NOT the interpolation
    flag, so the reference's `cv2.resize(mask, (64, 64), cv2.INTER_NEAREST)` (:190-193) runs the DEFAULT, INTER_LINEAR:
    source coordinate (d + 0.5) * scale - 0.5, floor / fraction, clamped to the edge; rows are interpolated horizontally
    first, then vertically, in the element type (double) with float coefficients.
"""
import math

import numpy as np
import torch
from PIL import Image


class InterpolationMode:
    NEAREST, BILINEAR, BICUBIC = 'nearest', 'bilinear', 'bicubic'


_PIL_MODE = {InterpolationMode.NEAREST: Image.NEAREST, InterpolationMode.BILINEAR: Image.BILINEAR,
             InterpolationMode.BICUBIC: Image.BICUBIC, 0: Image.NEAREST, 2: Image.BILINEAR, 3: Image.BICUBIC}


def _output_size(h, w, size, max_size):
    if isinstance(size, (list, tuple)) and len(size) == 2:
        return int(size[0]), int(size[1])
    if isinstance(size, (list, tuple)):
        size = size[0]
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    if max_size is not None:
        if max_size <= size:
            raise ValueError('max_size must be strictly greater than the requested size for the smaller edge')
        if new_long > max_size:
            new_short, new_long = int(max_size * new_short / new_long), max_size
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    return new_h, new_w


def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
    w, h = img.size
    nh, nw = _output_size(h, w, size, max_size)
    return img.resize((nw, nh), _PIL_MODE[interpolation])


def crop(img, top, left, height, width):
    return img.crop((left, top, left + width, top + height))


def hflip(img):
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def to_tensor(pic):
    a = np.array(pic)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


class _Transform(torch.nn.Module):
    pass


class Resize(_Transform):

    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
        super().__init__()
        self.size, self.interpolation, self.max_size = size, interpolation, max_size

    def forward(self, img):
        return resize(img, self.size, self.interpolation, self.max_size)


class RandomCrop(_Transform):

    def __init__(self, size):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def forward(self, img):
        w, h = img.size
        th, tw = self.size
        if h < th or w < tw:
            raise ValueError(f'Required crop size {(th, tw)} is larger than input image size {(h, w)}')
        if w == tw and h == th:
            return crop(img, 0, 0, h, w)
        i = torch.randint(0, h - th + 1, size=(1, )).item()
        j = torch.randint(0, w - tw + 1, size=(1, )).item()
        return crop(img, i, j, th, tw)


class CenterCrop(_Transform):

    def __init__(self, size):
        super().__init__()
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def forward(self, img):
        w, h = img.size
        th, tw = self.size
        return crop(img, int(round((h - th) / 2.0)), int(round((w - tw) / 2.0)), th, tw)


class RandomHorizontalFlip(_Transform):

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, img):
        return hflip(img) if torch.rand(1) < self.p else img


class Normalize(_Transform):

    def __init__(self, mean, std, inplace=False):
        super().__init__()
        self.mean, self.std = mean, std

    def forward(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return t.sub(mean).div(std)


# ---- cv2 ---------------------------------------------------------------------------------------------------------------
INTER_NEAREST, INTER_LINEAR = 0, 1


def cv2_resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    """Plain loops (64 x 64 outputs in the transforms). `dst` swallows the reference's misplaced flag, as in OpenCV."""
    if interpolation != INTER_LINEAR:
        raise NotImplementedError('only the default INTER_LINEAR is restated')
    src = np.asarray(src, dtype=np.float64)
    ow, oh = dsize
    ih, iw = src.shape[:2]

    def taps(n_out, n_in):
        out = []
        scale = n_in / n_out
        for d in range(n_out):
            f = (d + 0.5) * scale - 0.5
            s = math.floor(f)
            f = np.float32(f - s)
            if s < 0:
                s, f = 0, np.float32(0)
            if s >= n_in - 1:
                s, f = n_in - 1, np.float32(0)
            out.append((s, min(s + 1, n_in - 1), float(np.float32(1) - f), float(f)))
        return out

    xt, yt = taps(ow, iw), taps(oh, ih)
    rows = np.empty((ih, ow), dtype=np.float64)
    for x, (x0, x1, a0, a1) in enumerate(xt):
        rows[:, x] = src[:, x0] * a0 + src[:, x1] * a1
    out = np.empty((oh, ow), dtype=np.float64)
    for y, (y0, y1, b0, b1) in enumerate(yt):
        out[y] = rows[y0] * b0 + rows[y1] * b1
    return out
