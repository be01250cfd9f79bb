"""CLIP image transform on the host (PIL), as the reference's DataLoader workers run it:
Resize(bicubic) -> CenterCrop -> RGB -> ToTensor -> Normalize (OpenAI mean/std).

torchvision is not a dependency; the same operations are done with Pillow + torch.  Exposed as a
Compose-like object with ``.transforms`` so ``dataset.transform = preprocess`` and
``self.transforms.transform(image)`` (oadp/oake/globals.py:32, blocks.py:81) keep working.
"""
from __future__ import annotations

import numpy as np
import PIL
import PIL.Image
import torch

# The device resampler (csrc/resample.hip) restates Pillow's ImagingResample — fixed-point coefficients, uint8
# intermediate, and the undocumented choice of which pass runs first for sources > 100x taller than wide — and is
# pinned bit for bit against THIS Pillow release (tests/test_resample.py, oracle/resample_ref.py).  With another
# release the device path is still a correct antialiased bicubic resampler, but "bit-identical to the host PIL
# path" is only known for the release below: `check_pillow_version` says so once instead of staying silent.
PILLOW_PINNED = (12, 2)
_pillow_warned = False


def check_pillow_version(version: str | None = None) -> bool:
    """True if the installed Pillow is the release the device resampler was pinned against; otherwise warn
    (once per process) and return False.  Called when a dataset enables ``device_preprocess``."""
    global _pillow_warned
    import warnings
    version = version or PIL.__version__
    try:
        got = tuple(int(x) for x in version.split('.')[:2])
    except ValueError:
        got = ()
    if got == PILLOW_PINNED:
        return True
    if not _pillow_warned:
        _pillow_warned = True
        warnings.warn(
            f'Pillow {version} is installed; the GPU crop/resize path is pinned bit-exactly against Pillow '
            f'{PILLOW_PINNED[0]}.{PILLOW_PINNED[1]} (fixed-point bicubic + pass order). Features may differ in the '
            'last bit from the host PIL path under this release; run tests/test_resample.py to re-pin, or use '
            'device_preprocess=False for the reference\'s own PIL transform.', RuntimeWarning, stacklevel=2)
    return False


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def resize_short_side(image: PIL.Image.Image, size: int) -> PIL.Image.Image:
    """torchvision.transforms.Resize(size, BICUBIC) on a PIL image: short side -> size, long side
    int(size * long / short) (torchvision 0.13 ``_compute_resized_output_size``)."""
    w, h = image.size
    if (w <= h and w == size) or (h <= w and h == size):
        return image
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    return image.resize((ow, oh), PIL.Image.BICUBIC)


def center_crop(image: PIL.Image.Image, size: int) -> PIL.Image.Image:
    w, h = image.size
    left = int(round((w - size) / 2.0))
    top = int(round((h - size) / 2.0))
    return image.crop((left, top, left + size, top + size))


def to_tensor(image: PIL.Image.Image) -> torch.Tensor:
    """torchvision ToTensor: uint8 HWC -> float32 CHW / 255."""
    arr = np.asarray(image.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255)


def normalize(t: torch.Tensor) -> torch.Tensor:
    mean = torch.tensor(CLIP_MEAN, dtype=torch.float32).view(3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=torch.float32).view(3, 1, 1)
    return t.sub(mean).div(std)


class Preprocess:
    """``squash=False``: Resize(n) + CenterCrop(n) (OpenAI ``_transform``; the centre-square crop
    implied by block 0's bbox rule, oadp/oake/blocks.py:97-101).  ``squash=True``: Resize((n, n))
    without cropping — our reading of ``clip.load_default(True)`` used by globals (the fork is
    un-vendored; SURVEY.md Appendix D.1 — unpinned)."""

    def __init__(self, n_px: int = 224, squash: bool = False) -> None:
        self.n_px = n_px
        self.squash = squash

    @property
    def transform(self) -> 'Preprocess':  # ``dataset.transforms.transform(image)``
        return self

    def __call__(self, image: PIL.Image.Image) -> torch.Tensor:
        image = image.convert('RGB')
        if self.squash:
            if image.size != (self.n_px, self.n_px):
                image = image.resize((self.n_px, self.n_px), PIL.Image.BICUBIC)
        else:
            image = center_crop(resize_short_side(image, self.n_px), self.n_px)
        return normalize(to_tensor(image))
