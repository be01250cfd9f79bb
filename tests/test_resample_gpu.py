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
