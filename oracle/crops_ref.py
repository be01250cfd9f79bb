"""CPU restatement of OAKE's crop index math.  Test infrastructure only (see oracle/__init__.py).

Pure Python / numpy, written independently of ``oadp_amd.oake`` so the two can check each other,
and pinned against tests/golden/*.json — outputs of the reference's own functions run under import
stubs (tools/gen_golden.py).
"""
from __future__ import annotations

import math

import numpy as np
import PIL.Image

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)


# ---- blocks (reference oadp/oake/blocks.py:40-109) ------------------------------------------------
def partition(length: int, r: int = 224, s: int = 112) -> list[int]:
    """blocks.py:40-52: [] below r, [0] at r, else ceil((length-r)/s) near-equal integer steps."""
    if length < r:
        return []
    if length == r:
        return [0]
    n = (length - r - 1) // s + 1
    q, rem = divmod(length - r, n)
    steps = [q + 1] * rem + [q] * (n - rem)
    return [0] + list(np.cumsum(steps).tolist())


def pyramid_sizes(w: int, h: int, r: int = 224, rescale: float = 1.5) -> list[tuple[int, int, float]]:
    """blocks.py:54-77: (w, h, scale) per level until a side drops below r."""
    out, scale = [], 1.0
    while w >= r and h >= r:
        out.append((w, h, scale))
        w, h = int(w / rescale), int(h / rescale)
        scale *= rescale
    return out


def partitions(w: int, h: int, r: int = 224, s: int = 112, rescale: float = 1.5):
    """[(level_w, level_h, scale, x, y)] — x-major product of the two axis partitions per level."""
    tiles = []
    for lw, lh, scale in pyramid_sizes(w, h, r, rescale):
        for x in partition(lw, r, s):
            for y in partition(lh, r, s):
                tiles.append((lw, lh, scale, x, y))
    return tiles


def block_bbox(scale: float, x: int, y: int, r: int = 224) -> tuple[float, float, float, float]:
    """blocks.py:83-87."""
    return (x * scale, y * scale, x * scale + r * scale, y * scale + r * scale)


def block0_bbox(w: int, h: int) -> tuple[float, float, float, float]:
    """blocks.py:97-101 — literally (x, y, side, side)."""
    return ((w - h) / 2, 0, h, h) if w > h else (0, (h - w) / 2, w, w)


def all_block_bboxes(w: int, h: int) -> np.ndarray:
    rows = [block0_bbox(w, h)] + [block_bbox(sc, x, y) for _, _, sc, x, y in partitions(w, h)]
    return np.asarray(rows, dtype=np.float32)


# ---- objects (reference oadp/oake/objects.py:76-186) ----------------------------------------------
def keep_min_wh(boxes: np.ndarray, min_wh=(4, 4)) -> np.ndarray:
    boxes = np.asarray(boxes, dtype=np.float32)
    return ((boxes[:, 2] - boxes[:, 0]) >= min_wh[0]) & ((boxes[:, 3] - boxes[:, 1]) >= min_wh[1])


def expand_adaptive(boxes: np.ndarray, image_wh) -> np.ndarray:
    """objects.py:92-114 (ADAPTIVE), float32 arithmetic like torch."""
    b = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    wh_img = np.asarray(image_wh, dtype=np.float32)
    wh = b[:, 2:] - b[:, :2]
    length = np.sqrt((wh[:, 0] * wh[:, 1]) * np.float32(8))[:, None].astype(np.float32)
    center = (b[:, :2] + b[:, 2:]) / np.float32(2)
    # todd.BBoxesCXCYWH -> lt/rb
    lt = center - length / np.float32(2)
    rb = center + length / np.float32(2)
    offset = np.zeros_like(lt)
    offset = np.where(lt >= 0, offset, -lt)
    offset = np.where(rb <= wh_img, offset, wh_img - rb)
    offset = np.where((rb - lt) <= wh_img, offset, np.float32(0))
    return np.concatenate([lt + offset, rb + offset], axis=1).astype(np.float32)


def object_mask(foreground, obj, grid: int = 14) -> np.ndarray:
    """objects.py:129-155: nearest-neighbour resample of the (h x w) background mask to grid^2.
    torch 'nearest': src index = floor(dst * in / out) computed in float32."""
    w = len(np.arange(obj[2] - obj[0]))  # float arange => ceil length
    h = len(np.arange(obj[3] - obj[1]))
    xs, ys = np.arange(w), np.arange(h)
    in_x = (foreground[0] <= xs) & (xs <= foreground[2])
    in_y = (foreground[1] <= ys) & (ys <= foreground[3])
    mask = ~(in_y[:, None] & in_x[None, :])

    def src(n_in: int) -> np.ndarray:
        scale = np.float32(n_in) / np.float32(grid)
        idx = np.floor(np.arange(grid, dtype=np.float32) * scale).astype(np.int64)
        return np.minimum(idx, n_in - 1)

    return mask[src(h)][:, src(w)].astype(np.uint8)


def pil_crop_box(box) -> tuple[int, int, int, int]:
    """PIL.Image.crop rounds every coordinate with Python round() (banker's)."""
    return tuple(int(round(float(v))) for v in box)


# ---- preprocess (CLIP transform; torchvision is un-vendored -> restated with Pillow) --------------
def preprocess_ref(image: PIL.Image.Image, n: int = 224) -> np.ndarray:
    image = image.convert('RGB')
    w, h = image.size
    if not ((w <= h and w == n) or (h <= w and h == n)):
        if w < h:
            image = image.resize((n, int(n * h / w)), PIL.Image.BICUBIC)
        else:
            image = image.resize((int(n * w / h), n), PIL.Image.BICUBIC)
    w, h = image.size
    left, top = int(round((w - n) / 2.0)), int(round((h - n) / 2.0))
    image = image.crop((left, top, left + n, top + n))
    a = np.asarray(image, dtype=np.uint8).astype(np.float32) / np.float32(255)
    a = (a - CLIP_MEAN) / CLIP_STD
    return np.ascontiguousarray(a.transpose(2, 0, 1))
