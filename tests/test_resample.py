"""Pin the resampler restatement (oracle/resample_ref.py) against Pillow itself, bit for bit."""
import numpy as np
import PIL.Image
import pytest

from oracle import resample_ref


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    # add smooth structure + saturated regions so clipping and antialiasing both matter
    yy, xx = np.mgrid[0:h, 0:w]
    base[..., 0] = np.clip(base[..., 0].astype(int) // 2 + (xx * 255 // max(w - 1, 1)) // 2, 0, 255)
    base[:h // 5, :w // 5] = 255
    base[-h // 6:, -w // 6:] = 0
    return base


@pytest.mark.parametrize('w,h,ow,oh', [(640, 480, 426, 320), (426, 320, 284, 213), (100, 80, 224, 224),
                                       (37, 53, 224, 224), (500, 375, 298, 224), (1700, 1134, 1133, 756),
                                       (224, 300, 224, 224), (301, 299, 225, 224), (9, 7, 224, 224)])
def test_resize_matches_pillow(w, h, ow, oh):
    img = _img(w, h, w * 31 + h)
    ref = np.asarray(PIL.Image.fromarray(img).resize((ow, oh), PIL.Image.BICUBIC))
    got = resample_ref.resize_ref(img, ow, oh)
    assert np.array_equal(got, ref)


def test_default_resize_filter_is_bicubic():
    img = _img(90, 60, 1)
    a = np.asarray(PIL.Image.fromarray(img).resize((60, 40)))
    assert np.array_equal(a, resample_ref.resize_ref(img, 60, 40))


def test_crop_matches_pillow():
    img = _img(120, 90, 5)
    pil = PIL.Image.fromarray(img)
    for box in [(0.5, 1.5, 40.5, 60.5), (-10.2, -5.7, 50.1, 44.4), (100, 70, 140, 110), (2.5, 3.5, 4.5, 6.5)]:
        assert np.array_equal(resample_ref.crop_ref(img, box), np.asarray(pil.crop(box)))


def test_pass_order_of_very_narrow_sources():
    """Pillow resamples horizontally first — except a source more than 100 times taller than wide whose
    vertical pass reduces, which goes vertically first (the uint8 intermediate makes the order visible).  The
    rule is not documented; this pins `resample_ref.vertical_first` on both sides of every edge of it."""
    cases = [(2, 2000, 224, 224), (19, 2000, 224, 224), (20, 2000, 224, 224), (2, 201, 224, 100), (2, 200, 224, 100),
             (5, 501, 3, 100), (5, 500, 3, 100), (2, 300, 224, 299), (2, 300, 224, 301), (2, 300, 224, 600),
             (29, 3000, 448, 448), (30, 3000, 448, 448), (2000, 2, 224, 224), (300, 2, 100, 224), (1, 500, 7, 50)]
    seen = set()
    for w, h, ow, oh in cases:
        img = np.random.default_rng(w * 7 + h).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(PIL.Image.fromarray(img).resize((ow, oh), PIL.Image.BICUBIC))
        assert np.array_equal(resample_ref.resize_ref(img, ow, oh), ref), (w, h, ow, oh)
        seen.add(resample_ref.vertical_first(w, h, oh))
    assert seen == {True, False}
    rng = np.random.default_rng(3)
    for _ in range(60):  # random narrow shapes around the threshold
        w = int(rng.choice([2, 3, 4, 5, 7, 9, 12]))
        h = int(rng.integers(50, 1500))
        ow, oh = int(rng.choice([2, 3, 5, 8, 50, 224])), int(rng.integers(2, 1800))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = np.asarray(PIL.Image.fromarray(img).resize((ow, oh), PIL.Image.BICUBIC))
        assert np.array_equal(resample_ref.resize_ref(img, ow, oh), ref), (w, h, ow, oh)


def test_fixed_point_coefficients_fit_the_24_bit_multiply():
    """Advisor r05: the device resampler multiplies pixel x coefficient with 24-bit integer multiplies
    (csrc/resample.hip::tap).  Pillow's normalised bicubic weights EXCEED 1 where a window is cut at an image border
    (the negative lobe of one side is clipped away), so the bound the kernel relies on is |k| < 2^23, not |w| <= 1 —
    checked here over up- and down-scaling windows incl. every border window, on the oracle's coefficients (pinned
    bit-exactly to Pillow above; the kernel computes the same integers and clamps at the bound)."""
    from oracle.resample_ref import PRECISION_BITS, coeffs
    worst = 0
    for in_size, out_size in [(7, 224), (32, 224), (224, 224 * 3), (640, 224), (1700, 224), (3, 2), (2, 3), (1, 224),
                              (5, 4), (500, 499), (33, 224), (480, 320), (1134, 756), (13, 5)]:
        _, kk = coeffs(in_size, out_size)
        worst = max(worst, int(np.abs(kk).max()))
    assert (1 << PRECISION_BITS) < worst < (1 << 23), worst  # above 1.0 at a clipped border, far inside 24 signed bits
