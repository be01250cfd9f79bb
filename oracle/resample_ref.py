"""numpy restatement of Pillow's 8-bit antialiased bicubic resampler (src/libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc).
Test infrastructure only.  PARITY PIN: Pillow itself is importable here and on the GPU box, so this
restatement (and through it the device kernels of csrc/resample.hip, which follow the same steps) is
checked bit-for-bit against ``PIL.Image.resize`` in tests/test_resample.py."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: np.ndarray) -> np.ndarray:
    a = -0.5
    x = np.abs(x)
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def coeffs(in_size: int, out_size: int):
    """-> (bounds [out,2] int, kk [out,ksize] int32) exactly as Pillow computes them."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        w = _bicubic((x + xmin - center + 0.5) * ss)
        ww = 0.0
        for v in w:  # sequential sum, like the C loop
            ww += v
        if ww != 0.0:
            w = w / ww
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)), (0.5 + w * (1 << PRECISION_BITS)))
        kk[xx, :xmax] = np.trunc(q).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One resampling pass along `axis` of an HxWx3 uint8 image."""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    bounds, kk = coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :cnt].astype(np.int64), src[xmin:xmin + cnt], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def vertical_first(in_w: int, in_h: int, out_h: int) -> bool:
    """Pillow's pass order.  Horizontal first — except for a source more than 100 times taller than wide whose
    vertical pass reduces (Pillow 12.2; not in the documentation: pinned empirically against PIL.Image.resize
    over the threshold and 377 random narrow shapes, tests/test_resample.py)."""
    return in_h > 100 * in_w and out_h < in_h


def resize_ref(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.fromarray(img).resize((out_w, out_h), BICUBIC): two passes with a uint8 intermediate."""
    if vertical_first(img.shape[1], img.shape[0], out_h):
        return _pass(_pass(img, out_h, 0), out_w, 1)
    return _pass(_pass(img, out_w, 1), out_h, 0)


def crop_ref(img: np.ndarray, box) -> np.ndarray:
    """PIL Image.crop: round-half-even coordinates, zero fill outside."""
    x0, y0, x1, y1 = (int(round(float(v))) for v in box)
    h, w = img.shape[:2]
    out = np.zeros((y1 - y0, x1 - x0, 3), dtype=np.uint8)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out
