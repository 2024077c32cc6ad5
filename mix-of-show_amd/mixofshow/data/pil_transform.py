"""Image / caption transforms of the ED-LoRA data pipeline (reference mixofshow/data/pil_transform.py:17-364), rebuilt on
PIL + numpy + torch only (torchvision and cv2 are not dependencies of this framework).

Semantics that matter for identical training data, and where they come from:
  * the reference resizes PIL images through torchvision's functional API, which for PIL inputs IS `Image.resize` with
    the size rule of `_compute_resized_output_size`: shorter edge -> `size`, longer edge -> int(size * long / short), and
    with `max_size` the pair is rescaled so the longer edge equals it (:134,162,206,218);
  * `RandomCrop` draws its offsets with torch.randint (row first, then column; no draw when the sizes already match),
    `PairRandomCrop` and the placement offsets with Python's `random` (:55-56, :141-170) — the RNG streams are kept apart
    exactly like that, so a seeded run consumes both generators in the reference order;
  * the 1/8-resolution masks come from `cv2.resize(mask, (s//8, s//8), cv2.INTER_NEAREST)` (:190-193): the third positional
    argument of cv2.resize is `dst`, not the interpolation flag, so the reference actually gets cv2's DEFAULT, bilinear
    without anti-aliasing (half-pixel centres). `_cv2_resize_linear` restates that: at the 8x reduction used here every
    output cell is the mean of the 2x2 source pixels around its centre — masks have fractional values on their borders.
Every transform takes and returns `(img, kwargs)` when it handles the side information (mask / img_mask / prompts), or
just the image otherwise; `PairCompose` dispatches on the call signature like the reference (:101-112).
"""
import inspect
import math
import random
from copy import deepcopy

import numpy as np
import torch
from PIL import Image

_REGISTRY = {}


def register(cls):
    _REGISTRY[cls.__name__] = cls
    return cls


def build_transform(opt):
    opt = deepcopy(opt)
    kind = opt.pop('type')
    if kind not in _REGISTRY:
        raise KeyError(f'unknown transform {kind!r}; available: {sorted(_REGISTRY)}')
    return _REGISTRY[kind](**opt)


# ---- geometry helpers ---------------------------------------------------------------------------------------------
def _resized_size(w, h, size, max_size=None):
    """torchvision `_compute_resized_output_size` for an int `size` (returns (w, h))."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_short, new_long) if w <= h else (new_long, new_short)


def resize(img, size, max_size=None, resample=Image.BILINEAR):
    if isinstance(size, (tuple, list)):
        h, w = size
        return img.resize((w, h), resample)
    w, h = img.size
    nw, nh = _resized_size(w, h, size, max_size)
    return img if (nw, nh) == (w, h) else img.resize((nw, nh), resample)


def crop(img, top, left, height, width):
    return img.crop((left, top, left + width, top + height))


def _torch_random_crop_params(w, h, tw, th):
    if h < th or w < tw:
        raise ValueError(f'Required crop size {(th, tw)} is larger than input image size {(h, w)}')
    if w == tw and h == th:
        return 0, 0
    i = int(torch.randint(0, h - th + 1, size=(1, )).item())
    j = int(torch.randint(0, w - tw + 1, size=(1, )).item())
    return i, j


def _cv2_resize_linear(a, out_w, out_h):
    """cv2.resize(a, (out_w, out_h)) with its default INTER_LINEAR (half-pixel centres, edge replicate, no anti-aliasing)."""
    a = np.asarray(a, dtype=np.float64)
    in_h, in_w = a.shape[:2]

    def taps(n_out, n_in):
        x = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        x0 = np.floor(x).astype(np.int64)
        f = x - x0
        lo, hi = np.clip(x0, 0, n_in - 1), np.clip(x0 + 1, 0, n_in - 1)
        return lo, hi, f

    ylo, yhi, fy = taps(out_h, in_h)
    xlo, xhi, fx = taps(out_w, in_w)
    top = a[ylo][:, xlo] * (1 - fx) + a[ylo][:, xhi] * fx
    bot = a[yhi][:, xlo] * (1 - fx) + a[yhi][:, xhi] * fx
    return top * (1 - fy)[:, None] + bot * fy[:, None]


def _place_on_canvas(img, kwargs, size):
    """Random placement of the (<= size) image on a zero canvas + the image-validity mask, then the 1/8-resolution masks
    (reference :164-193 / :224-252). Offsets: random.randint, row first. The reference hard-wires 512 in the offset range
    (:170) and therefore only works at size 512; `size` is used here, identical at 512."""
    new_w, new_h = img.size
    arr = np.array(img)
    mask = np.array(kwargs['mask']) / 255 if 'mask' in kwargs else None
    start_y = random.randint(0, size - new_h)
    start_x = random.randint(0, size - new_w)
    canvas = np.zeros((size, size, 3), dtype=np.uint8)
    canvas[start_y:start_y + new_h, start_x:start_x + new_w, :] = arr
    img_mask = np.zeros((size, size))
    img_mask[start_y:start_y + new_h, start_x:start_x + new_w] = 1
    if mask is not None:
        full = np.zeros((size, size))
        full[start_y:start_y + new_h, start_x:start_x + new_w] = mask
        kwargs['mask'] = torch.from_numpy(_cv2_resize_linear(full, size // 8, size // 8))
    kwargs['img_mask'] = torch.from_numpy(_cv2_resize_linear(img_mask, size // 8, size // 8))
    return Image.fromarray(canvas), kwargs


# ---- plain image transforms (torchvision names the YAML recipes use) ----------------------------------------------------
@register
class Resize:

    def __init__(self, size, interpolation=Image.BILINEAR):
        self.size, self.interpolation = size, interpolation

    def forward(self, img):
        return resize(img, self.size, resample=self.interpolation)


@register
class BILINEARResize(Resize):

    def __init__(self, size):
        super().__init__(size, Image.BILINEAR)


@register
class CenterCrop:

    def __init__(self, size):
        self.th, self.tw = (size, size) if isinstance(size, int) else size

    def forward(self, img):
        w, h = img.size
        return crop(img, int(round((h - self.th) / 2.0)), int(round((w - self.tw) / 2.0)), self.th, self.tw)


@register
class RandomCrop:

    def __init__(self, size):
        self.th, self.tw = (size, size) if isinstance(size, int) else size

    def forward(self, img):
        w, h = img.size
        i, j = _torch_random_crop_params(w, h, self.tw, self.th)
        return crop(img, i, j, self.th, self.tw)


@register
class RandomHorizontalFlip:

    def __init__(self, p=0.5):
        self.p = p

    def forward(self, img):
        return img.transpose(Image.FLIP_LEFT_RIGHT) if torch.rand(1) < self.p else img


@register
class ToTensor:
    """PIL (H, W, C) uint8 -> float (C, H, W) in [0, 1] (torchvision to_tensor)."""

    def forward(self, pic):
        a = np.array(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(a.transpose(2, 0, 1).copy())
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()


@register
class Normalize:

    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def forward(self, t):
        mean = torch.tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std


# ---- paired (image + mask / prompts) transforms -----------------------------------------------------------------------
@register
class PairRandomCrop:
    """Same window on image and mask; offsets from Python's `random` (x first, then y; reference :55-56)."""

    def __init__(self, size):
        self.height, self.width = (size, size) if isinstance(size, int) else size

    def forward(self, img, **kwargs):
        w, h = img.size
        mw, mh = kwargs['mask'].size
        assert h >= self.height and h == mh and w >= self.width and w == mw
        x = random.randint(0, w - self.width)
        y = random.randint(0, h - self.height)
        kwargs['mask'] = crop(kwargs['mask'], y, x, self.height, self.width)
        return crop(img, y, x, self.height, self.width), kwargs


@register
class PairRandomHorizontalFlip:

    def __init__(self, p=0.5):
        self.p = p

    def forward(self, img, **kwargs):
        if torch.rand(1) < self.p:
            kwargs['mask'] = kwargs['mask'].transpose(Image.FLIP_LEFT_RIGHT)
            return img.transpose(Image.FLIP_LEFT_RIGHT), kwargs
        return img, kwargs


@register
class PairResize:

    def __init__(self, size):
        self.size = size

    def forward(self, img, **kwargs):
        kwargs['mask'] = resize(kwargs['mask'], self.size)
        return resize(img, self.size), kwargs


class PairCompose:
    """Transforms whose `forward` takes only the image get the image; the others get (img, **kwargs) (reference :101-112)."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, img, **kwargs):
        for t in self.transforms:
            if len(inspect.signature(t.forward).parameters) == 1:
                img = t.forward(img)
            else:
                img, kwargs = t.forward(img, **kwargs)
        return img, kwargs


@register
class HumanResizeCropFinalV3:
    """reference :125-195 — short edge to `size`; with probability crop_p a random crop (portrait: keep the top `w + r`
    rows; landscape / square: a size x size window); long edge to `size`; random placement on a size x size canvas."""

    def __init__(self, size, crop_p=0.5):
        self.size, self.crop_p = size, crop_p
        self._crop = RandomCrop(size)
        self._pair_crop = PairRandomCrop(size)

    def forward(self, img, **kwargs):
        has_mask = 'mask' in kwargs
        img = resize(img, self.size)
        if has_mask:
            kwargs['mask'] = resize(kwargs['mask'], self.size)
        w, h = img.size
        if random.random() < self.crop_p:
            if h > w:
                extra = random.randint(0, h - w)
                img = crop(img, 0, 0, w + extra, w)
                if has_mask:
                    kwargs['mask'] = crop(kwargs['mask'], 0, 0, w + extra, w)
            elif has_mask:
                img, kwargs = self._pair_crop.forward(img, **kwargs)
            else:
                img = self._crop.forward(img)
        img = resize(img, self.size - 1, max_size=self.size)
        if has_mask:
            kwargs['mask'] = resize(kwargs['mask'], self.size - 1, max_size=self.size)
        return _place_on_canvas(img, kwargs, self.size)


@register
class ResizeFillMaskNew:
    """reference :198-254 — short edge to `size`; random size x size crop (prob. crop_p) or long edge to `size`; random
    isotropic rescale by U(scale_ratio) (mask: nearest); random placement on the canvas."""

    def __init__(self, size, crop_p, scale_ratio):
        self.size, self.crop_p, self.scale_ratio = size, crop_p, scale_ratio
        self._crop = RandomCrop(size)
        self._pair_crop = PairRandomCrop(size)

    def forward(self, img, **kwargs):
        has_mask = 'mask' in kwargs
        img = resize(img, self.size)
        if has_mask:
            kwargs['mask'] = resize(kwargs['mask'], self.size)
        if random.random() < self.crop_p:
            if has_mask:
                img, kwargs = self._pair_crop.forward(img, **kwargs)
            else:
                img = self._crop.forward(img)
        else:
            img = resize(img, self.size - 1, max_size=self.size)
            if has_mask:
                kwargs['mask'] = resize(kwargs['mask'], self.size - 1, max_size=self.size)
        w, h = img.size
        ratio = random.uniform(*self.scale_ratio)
        img = resize(img, (int(h * ratio), int(w * ratio)))
        if has_mask:
            kwargs['mask'] = resize(kwargs['mask'], (int(h * ratio), int(w * ratio)), resample=Image.NEAREST)
        return _place_on_canvas(img, kwargs, self.size)


@register
class ShuffleCaption:
    """Comma-separated caption: the first keep_token_num phrases stay, the rest is shuffled (reference :257-275)."""

    def __init__(self, keep_token_num):
        self.keep_token_num = keep_token_num

    def forward(self, img, **kwargs):
        phrases = [t.strip() for t in kwargs['prompts'].strip().strip().split(',')]
        fixed, flex = [], phrases
        if self.keep_token_num > 0:
            fixed, flex = phrases[:self.keep_token_num], phrases[self.keep_token_num:]
        random.shuffle(flex)
        kwargs['prompts'] = ', '.join(fixed + flex)
        return img, kwargs


# Caption templates of textual inversion as the reference lists them (:281-345) — the ORDER is part of the behaviour
# (random.choice indexes into it), so they are data reproduced as is.
_STYLE = ['a painting in the style of {}', 'a rendering in the style of {}', 'a cropped painting in the style of {}',
          'the painting in the style of {}', 'a clean painting in the style of {}', 'a dirty painting in the style of {}',
          'a dark painting in the style of {}', 'a picture in the style of {}', 'a cool painting in the style of {}',
          'a close-up painting in the style of {}', 'a bright painting in the style of {}',
          'a cropped painting in the style of {}', 'a good painting in the style of {}',
          'a close-up painting in the style of {}', 'a rendition in the style of {}', 'a nice painting in the style of {}',
          'a small painting in the style of {}', 'a weird painting in the style of {}', 'a large painting in the style of {}']
_OBJECT = ['a photo of a {}', 'a rendering of a {}', 'a cropped photo of the {}', 'the photo of a {}', 'a photo of a clean {}',
           'a photo of a dirty {}', 'a dark photo of the {}', 'a photo of my {}', 'a photo of the cool {}',
           'a close-up photo of a {}', 'a bright photo of the {}', 'a cropped photo of a {}', 'a photo of the {}',
           'a good photo of the {}', 'a photo of one {}', 'a close-up photo of the {}', 'a rendition of the {}',
           'a photo of the clean {}', 'a rendition of a {}', 'a photo of a nice {}', 'a good photo of a {}',
           'a photo of the nice {}', 'a photo of the small {}', 'a photo of the weird {}', 'a photo of the large {}',
           'a photo of a cool {}', 'a photo of a small {}']
_HUMAN = ['a photo of a {}', 'a photo of one {}', 'a photo of the {}', 'the photo of a {}', 'a rendering of a {}',
          'a rendition of the {}', 'a rendition of a {}', 'a cropped photo of the {}', 'a cropped photo of a {}',
          'a bad photo of the {}', 'a bad photo of a {}', 'a photo of a weird {}', 'a weird photo of a {}',
          'a bright photo of the {}', 'a good photo of the {}', 'a photo of a nice {}', 'a good photo of a {}',
          'a photo of a cool {}', 'a bright photo of the {}']


@register
class EnhanceText:
    """The caption becomes a random template around the concept token(s) (reference :278-364)."""

    def __init__(self, enhance_type='object'):
        try:
            self.templates = {'object': _OBJECT, 'style': _STYLE, 'human': _HUMAN}[enhance_type]
        except KeyError:
            raise NotImplementedError(enhance_type)

    def forward(self, img, **kwargs):
        kwargs['prompts'] = random.choice(self.templates).format(kwargs['prompts'].strip())
        return img, kwargs
