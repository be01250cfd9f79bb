"""Device crop + antialiased bicubic resize + ToTensor + Normalize vs Pillow itself: bit-exact."""
import numpy as np
import PIL.Image
import pytest
import torch

from oadp_amd import clip
from oadp_amd.clip.preprocess import Preprocess
from oadp_amd.weights import synthetic_state_dict

from . import _synth

pytestmark = pytest.mark.gpu


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    a[..., 1] = np.clip(a[..., 1].astype(int) // 2 + (yy * 255 // max(h - 1, 1)) // 2, 0, 255)
    a[:h // 4, :w // 4] = 255
    return a


@pytest.fixture(scope='module')
def visual(cuda):
    model, _ = clip.load(synthetic_state_dict(**_synth.TINY), max_batch=4)
    return model.visual


@pytest.mark.parametrize('w,h', [(640, 480), (500, 375), (213, 300), (1700, 1134)])
def test_pyramid_resize_matches_pillow(visual, cuda, w, h):
    img = _img(w, h, w + h)
    level = torch.from_numpy(img).to(cuda)
    pil = PIL.Image.fromarray(img)
    for _ in range(3):  # three pyramid steps, each from the previous level's uint8 image
        ww, hh = pil.size
        size = (int(ww / 1.5), int(hh / 1.5))
        pil = pil.resize(size)
        level = visual.resize_u8(level, size)
        assert np.array_equal(level.cpu().numpy(), np.asarray(pil))


def test_object_crops_match_pillow(visual, cuda):
    """preprocess(image.crop(box)) for proposal-style boxes: fractional coordinates, boxes sticking
    out of the image (zero fill), tiny (up-sampled) and large (anti-aliased down-sampled) crops."""
    img = _img(640, 480, 3)
    pil = PIL.Image.fromarray(img)
    pre = Preprocess(224, squash=False)
    rng = np.random.default_rng(0)
    boxes = [(10.4, 20.6, 130.4, 140.6), (-30.5, -12.5, 193.5, 211.5), (500.2, 300.7, 700.2, 500.7),
             (0, 0, 640, 480), (100, 50, 110, 60), (3.5, 4.5, 227.5, 228.5), (0, 0, 224, 224),
             (50, 60, 151, 160), (50, 60, 150, 161)]
    for _ in range(40):
        side = float(np.exp(rng.uniform(np.log(6), np.log(700))))
        cx, cy = rng.uniform(0, 640), rng.uniform(0, 480)
        boxes.append((cx - side / 2, cy - side / 2, cx + side / 2, cy + side / 2))
    ref = torch.stack([pre(pil.crop(b)) for b in boxes])
    out = visual.crop_resize_normalize(torch.from_numpy(img).to(cuda), boxes)
    assert out.shape == ref.shape
    assert torch.equal(out.cpu(), ref)
    out16 = visual.crop_resize_normalize(torch.from_numpy(img).to(cuda), boxes, out_dtype=torch.float16)
    assert torch.equal(out16.cpu(), ref.half())


@pytest.mark.parametrize('w,h', [(640, 480), (480, 640), (224, 224), (1000, 300)])
def test_whole_image_preprocess_matches_pillow(visual, cuda, w, h):
    img = _img(w, h, 9)
    pil = PIL.Image.fromarray(img)
    d = torch.from_numpy(img).to(cuda)
    box = [(0, 0, w, h)]
    assert torch.equal(visual.crop_resize_normalize(d, box).cpu()[0], Preprocess(224, squash=False)(pil))
    assert torch.equal(visual.crop_resize_normalize(d, box, squash=True).cpu()[0],
                       Preprocess(224, squash=True)(pil))


def test_blocks_batch_matches_per_image_path(cuda):
    """oake_blocks_batch (pyramids + every block crop of a whole flush, batched by level across images) ==
    the image-by-image composition of crop_resize_normalize / crop_normalize / resize_u8 that is itself
    bit-identical to Pillow (tests above) — bit for bit, in both output types; 1700x1134 walks all five
    levels, 100x90 has block 0 only, 224x224 exactly one tile."""
    import itertools
    from oadp_amd.oake import blocks
    from oadp_amd import clip
    from oadp_amd.weights import synthetic_state_dict
    ds = blocks.Dataset.__new__(blocks.Dataset)
    ds._r, ds._s, ds._rescale = 224, 112, 1.5
    model, _ = clip.load(synthetic_state_dict(width=128, layers=1, heads=2, mlp_dim=256, embed_dim=64), max_batch=4)
    v = model.visual
    rng = np.random.default_rng(11)
    sizes = [(640, 480), (1700, 1134), (100, 90), (224, 224), (500, 375), (337, 336), (480, 640)]
    images = [torch.from_numpy(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).to(cuda) for w, h in sizes]
    for dt in (torch.float16, torch.float32):
        out, counts = v.blocks_batch(images, out_dtype=dt)
        assert counts == [1 + len(ds._level_tiles(w, h)) for w, h in sizes]
        i = 0
        for im, (w, h), k in zip(images, sizes, counts):
            ref = [v.crop_resize_normalize(im, [(0, 0, w, h)], out_dtype=dt)]
            level, lw, lh = im, w, h
            while True:
                tiles = list(itertools.product(ds._partition(lw), ds._partition(lh)))
                if not tiles:
                    break
                ref.append(v.crop_normalize(level, [(x, y, x + 224, y + 224) for x, y in tiles], out_dtype=dt))
                lw, lh = int(lw / 1.5), int(lh / 1.5)
                level = v.resize_u8(level, (lw, lh))
            ref = torch.cat(ref)
            assert ref.shape[0] == k
            assert torch.equal(out[i:i + k], ref), (w, h, dt)
            i += k
        assert i == out.shape[0]


@pytest.mark.parametrize('squash', [False, True])
def test_crop_boxes_of_every_kind_match_pillow(cuda, squash):
    """preprocess(image.crop(box)) on the device vs PIL for boxes inside the image, across every border, tiny,
    taller than wide and random, on sources from 37x53 to 2600x1900 — one batched call, bit for bit."""
    model, _ = clip.load(synthetic_state_dict(**_synth.TINY), max_batch=4)
    rng = np.random.default_rng(9)
    arrs = [_img(640, 480, 1), _img(333, 517, 2), _img(37, 53, 3), _img(1700, 1134, 4), _img(2600, 1900, 5)]
    boxes = []
    for a in arrs:
        h, w = a.shape[:2]
        bs = [(0, 0, w, h), (-20.5, -10.5, w * 0.6, h * 0.7), (w * 0.3, h * 0.2, w + 33.2, h + 5.5),
              (w * 0.45, h * 0.45, w * 0.45 + 4.2, h * 0.45 + 9.7), (w * 0.1, h * 0.05, w * 0.2, h * 0.95)]
        for _ in range(6):
            x1, y1 = rng.uniform(-30, w * 0.8), rng.uniform(-30, h * 0.8)
            bs.append((x1, y1, x1 + rng.uniform(4, w), y1 + rng.uniform(4, h)))
        boxes.append([tuple(float(v) for v in b) for b in bs])
    out = model.visual.crop_resize_normalize_batch([torch.from_numpy(a).to(cuda) for a in arrs], boxes,
                                                   squash=squash, out_dtype=torch.float32)
    host = Preprocess(224, squash=squash)
    i = 0
    for a, bs in zip(arrs, boxes):
        pil = PIL.Image.fromarray(a)
        for b in bs:
            assert torch.equal(out[i].cpu(), host(pil.crop(b))), (a.shape, b, squash)
            i += 1
