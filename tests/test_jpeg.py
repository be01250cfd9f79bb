"""JPEG decode: oracle pinned to Pillow, host Huffman pass pinned to the oracle (CPU),
device reconstruction pinned to Pillow (GPU)."""
import ctypes as C
import io

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import jpeg_ref


def _synth(h, w, kind, seed=0):
    rng = np.random.default_rng(seed + h * 131 + w)
    if kind == 'noise':
        return (rng.random((h, w, 3)) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.stack([(xx * 7 + yy * 3) % 256, (xx * 2 + yy * 5 + 40) % 256, (xx * yy) % 256], -1)
    return smooth.astype(np.uint8)


def _encode(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).save(b, 'JPEG', **kw)
    return b.getvalue()


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))


CASES = [(h, w, ss, q, kind)
         # (widths <= 4: libjpeg installs its fancy chroma upsampling only above two chroma samples per row)
         for (h, w) in [(8, 8), (1, 1), (7, 9), (17, 33), (37, 53), (64, 48), (33, 16), (15, 1), (14, 3), (53, 4),
                        (52, 2), (6, 5)]
         for ss in (0, 1, 2) for q in (30, 90) for kind in ('noise', 'grad')]


@pytest.mark.parametrize('h,w,ss,q,kind', CASES)
def test_oracle_matches_pillow(h, w, ss, q, kind):
    data = _encode(_synth(h, w, kind), quality=q, subsampling=ss)
    assert np.array_equal(jpeg_ref.decode(data), _pil(data))


def test_oracle_grayscale_optimized_restart():
    g = (np.random.default_rng(3).random((21, 35)) * 255).astype(np.uint8)
    data = _encode(g, quality=80)
    assert np.array_equal(jpeg_ref.decode(data), _pil(data))
    a = _synth(45, 70, 'grad')
    data = _encode(a, quality=85, optimize=True)  # image-specific Huffman tables
    assert np.array_equal(jpeg_ref.decode(data), _pil(data))
    data = _encode(a, quality=85, subsampling=2, restart_marker_blocks=3)
    assert b'\xff\xdd' in data  # DRI present
    assert np.array_equal(jpeg_ref.decode(data), _pil(data))


def test_oracle_progressive():
    data = _encode(_synth(16, 16, 'grad'), quality=80, progressive=True)
    with pytest.raises(NotImplementedError):
        jpeg_ref.decode(data)  # (opt-in: the baseline oracle stays the default)
    for (h, w, ss, q, kind) in CASES[::4]:
        data = _encode(_synth(h, w, kind), quality=q, subsampling=ss, progressive=True)
        assert np.array_equal(jpeg_ref.decode(data, progressive_ok=True), _pil(data))
    g = (np.random.default_rng(4).random((21, 35)) * 255).astype(np.uint8)
    data = _encode(g, quality=80, progressive=True)
    assert np.array_equal(jpeg_ref.decode(data, progressive_ok=True), _pil(data))


def _cmyk(a, **kw):
    b = io.BytesIO()
    Image.fromarray(a).convert('CMYK').save(b, 'JPEG', **kw)
    return b.getvalue()


def _host_coefs(lib, data):
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    total = C.c_size_t(0)
    rc = lib.oake_jpeg_entropy_decode(buf, len(data), None, 0, C.byref(total))
    assert rc == 0
    out = np.zeros(total.value, np.int16)
    rc = lib.oake_jpeg_entropy_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), out.size, C.byref(total))
    assert rc == 0
    return out


@pytest.mark.parametrize('h,w,ss,q,kind', CASES[::3])
def test_host_huffman_pass_matches_oracle(h, w, ss, q, kind):
    """The C++ entropy decoder of liboake_hip.so (no GPU involved) against the oracle's coefficients."""
    from oadp_amd import _lib
    lib = _lib.load()
    data = _encode(_synth(h, w, kind), quality=q, subsampling=ss)
    _, _, _, _, planes = jpeg_ref.parse(data)
    ref = np.concatenate([p.reshape(-1) for p in planes]).astype(np.int16)
    assert np.array_equal(_host_coefs(lib, data), ref)


def test_host_huffman_restart_optimized_info():
    from oadp_amd import _lib
    lib = _lib.load()
    a = _synth(45, 70, 'noise')
    for kw in (dict(quality=85, optimize=True), dict(quality=70, subsampling=2, restart_marker_blocks=2),
               dict(quality=95, subsampling=1, restart_marker_rows=1)):
        data = _encode(a, **kw)
        _, _, _, _, planes = jpeg_ref.parse(data)
        ref = np.concatenate([p.reshape(-1) for p in planes]).astype(np.int16)
        assert np.array_equal(_host_coefs(lib, data), ref)
        hh, ww, cc = C.c_int(), C.c_int(), C.c_int()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        assert lib.oake_jpeg_info(buf, len(data), C.byref(hh), C.byref(ww), C.byref(cc)) == 0
        assert (hh.value, ww.value, cc.value) == (45, 70, 3)
    for kw in (dict(quality=80, progressive=True), dict(quality=60, progressive=True, subsampling=2),
               dict(quality=92, progressive=True, subsampling=0, optimize=True)):
        data = _encode(a, **kw)  # progressive: several scans, spectral selection + successive approximation
        _, _, _, _, planes = jpeg_ref.parse(data, progressive_ok=True)
        ref = np.concatenate([p.reshape(-1) for p in planes]).astype(np.int16)
        assert np.array_equal(_host_coefs(lib, data), ref)
    cmyk = _cmyk(a, quality=80)
    buf = (C.c_uint8 * len(cmyk)).from_buffer_copy(cmyk)
    assert lib.oake_jpeg_info(buf, len(cmyk), None, None, None) == _lib.OAKE_ERR_UNSUPPORTED
    # fuzz: corrupted files must be rejected or decoded to *something*, never crash the process
    rng = np.random.default_rng(11)
    goods = [_encode(a, quality=85, subsampling=2), _encode(a, quality=75, subsampling=2, progressive=True)]
    for trial in range(600):
        good = goods[trial % 2]
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
        if trial % 3 == 0:
            bad = bad[:int(rng.integers(4, len(bad)))]
        buf = (C.c_uint8 * len(bad)).from_buffer_copy(bytes(bad))
        total = C.c_size_t(0)
        if lib.oake_jpeg_entropy_decode(buf, len(bad), None, 0, C.byref(total)) == 0 and total.value < 10 ** 7:
            out = np.zeros(total.value, np.int16)
            lib.oake_jpeg_entropy_decode(buf, len(bad), out.ctypes.data_as(C.c_void_p), out.size, C.byref(total))
    junk = bytes(100)
    buf = (C.c_uint8 * len(junk)).from_buffer_copy(junk)
    assert lib.oake_jpeg_info(buf, len(junk), None, None, None) == _lib.OAKE_ERR_INVALID


GPU_CASES = [(480, 640, 2, 85, 'grad'), (427, 640, 2, 75, 'noise'), (333, 500, 1, 90, 'grad'),
             (224, 224, 0, 95, 'noise'), (1, 1, 2, 50, 'grad'), (17, 33, 2, 60, 'noise'), (9, 7, 1, 60, 'noise'),
             (1134, 1700, 2, 80, 'grad'), (480, 1, 2, 28, 'noise'), (153, 4, 2, 34, 'grad'), (7, 4, 1, 93, 'noise'),
             (14, 3, 2, 76, 'grad'), (345, 2, 1, 91, 'noise'), (9, 5, 2, 70, 'noise')]


@pytest.mark.gpu
@pytest.mark.parametrize('h,w,ss,q,kind', GPU_CASES)
def test_device_decode_matches_pillow(cuda, h, w, ss, q, kind):
    """oake_decode_jpeg through the C ABI: every pixel equal to PIL.Image.open().convert('RGB')."""
    from oadp_amd import clip
    from oadp_amd.weights import synthetic_state_dict
    from tests._synth import TINY
    model, _ = clip.load(synthetic_state_dict(**TINY), max_batch=2)
    data = _encode(_synth(h, w, kind), quality=q, subsampling=ss)
    out = model.visual.decode_jpeg(data)
    assert out.dtype == torch.uint8 and out.shape == (h, w, 3) and out.is_cuda
    assert np.array_equal(out.cpu().numpy(), _pil(data))
    # the two halves separately (Huffman pass in a worker, reconstruction here)
    from oadp_amd import _lib
    coefs = torch.from_numpy(_host_coefs(_lib.load(), data))
    assert torch.equal(model.visual.decode_jpeg(data, coefs=coefs), out)


@pytest.mark.gpu
def test_device_decode_batch(cuda):
    """oake_decode_jpeg_batch: many files, Huffman passes on native threads; per-image status."""
    from oadp_amd import clip
    from oadp_amd.weights import synthetic_state_dict
    from tests._synth import TINY
    model, _ = clip.load(synthetic_state_dict(**TINY), max_batch=2)
    datas = [_encode(_synth(40 + 7 * i, 50 + 11 * i, 'grad' if i % 2 else 'noise', seed=i), quality=60 + 3 * i,
                     subsampling=i % 3) for i in range(12)]
    datas.insert(3, _encode(_synth(61, 47, 'grad'), quality=80, progressive=True, subsampling=2))
    datas.insert(5, _cmyk(_synth(32, 32, 'grad'), quality=80))  # unsupported -> None
    datas.insert(9, b'garbage')
    for threads in (1, 4, 32):
        outs = model.visual.decode_jpeg_batch(datas, threads=threads)
        assert outs[5] is None and outs[9] is None
        for d, o in zip(datas, outs):
            if o is not None:
                assert np.array_equal(o.cpu().numpy(), _pil(d))
    # file bytes as 1-D uint8 tensors (what the dataset hands over) are read in place; the images of a
    # call are views of one allocation, 256-byte aligned
    import torch
    mixed = [torch.frombuffer(bytearray(d), dtype=torch.uint8) if i % 2 else d for i, d in enumerate(datas)]
    outs = model.visual.decode_jpeg_batch(mixed, threads=8)
    for d, o in zip(datas, outs):
        if o is not None:
            assert o.data_ptr() % 256 == 0 and np.array_equal(o.cpu().numpy(), _pil(d))
    with pytest.raises(ValueError):
        model.visual.decode_jpeg_batch([torch.zeros(4, 4, dtype=torch.uint8)])
    assert model.visual.decode_jpeg_batch([b'garbage']) == [None] and model.visual.decode_jpeg_batch([]) == []


@pytest.mark.gpu
def test_crops_into_preallocated_batch(cuda):
    """out=: the sweep fills one [N,3,n,n] tensor per flush; same pixels as the allocating call."""
    import torch
    from oadp_amd import clip
    from oadp_amd.weights import synthetic_state_dict
    from tests._synth import TINY
    model, _ = clip.load(synthetic_state_dict(**TINY), max_batch=2)
    v = model.visual
    n = v.input_resolution
    img = torch.from_numpy(_synth(97, 143, 'noise', seed=5)).to(cuda)
    boxes = [(0, 0, 143, 97), (10.5, 3.2, 80.1, 60.0), (-5, -5, 40, 40)]
    want = v.crop_resize_normalize(img, boxes, out_dtype=torch.float16)
    batch = torch.zeros((5, 3, n, n), dtype=torch.float16, device=cuda)
    got = v.crop_resize_normalize(img, boxes, out_dtype=torch.float16, out=batch[1:4])
    assert got.data_ptr() == batch[1:4].data_ptr() and torch.equal(batch[1:4], want)
    assert not batch[0].any() and not batch[4].any()
    tiles = [(0, 0, n, n), (143 - n, 97 - n, 143, 97)] if n <= 97 else []
    if tiles:
        want = v.crop_normalize(img, tiles, out_dtype=torch.float16)
        v.crop_normalize(img, tiles, out_dtype=torch.float16, out=batch[0:2])
        assert torch.equal(batch[0:2], want)
    # several images in one native call == the per-image calls, image after image
    img2 = torch.from_numpy(_synth(60, 75, 'grad', seed=6)).to(cuda)
    boxes2 = [(0, 0, 75, 60)]
    got = v.crop_resize_normalize_batch([img, img2, img], [boxes, boxes2, []], out_dtype=torch.float16)
    want = torch.cat([v.crop_resize_normalize(img, boxes, out_dtype=torch.float16),
                      v.crop_resize_normalize(img2, boxes2, out_dtype=torch.float16)])
    assert torch.equal(got, want)
    assert v.crop_resize_normalize_batch([], []).shape == (0, 3, n, n)
    with pytest.raises(ValueError):
        v.crop_resize_normalize_batch([img], [boxes, boxes2])
    with pytest.raises(ValueError):
        v.crop_resize_normalize(img, boxes, out_dtype=torch.float16, out=batch[0:2])
    with pytest.raises(ValueError):
        v.crop_resize_normalize(img, boxes, out_dtype=torch.float32, out=batch[1:4])


@pytest.mark.gpu
def test_device_decode_gray_restart_errors(cuda):
    from oadp_amd import clip, _lib
    from oadp_amd.weights import synthetic_state_dict
    from tests._synth import TINY
    model, _ = clip.load(synthetic_state_dict(**TINY), max_batch=2)
    g = (np.random.default_rng(3).random((121, 235)) * 255).astype(np.uint8)
    for data in (_encode(g, quality=80), _encode(_synth(90, 130, 'noise'), quality=85, optimize=True),
                 _encode(_synth(90, 130, 'grad'), quality=70, subsampling=2, restart_marker_blocks=5)):
        assert np.array_equal(model.visual.decode_jpeg(data).cpu().numpy(), _pil(data))
    for data in (_encode(_synth(90, 130, 'grad'), quality=80, progressive=True, subsampling=2),
                 _encode(_synth(33, 70, 'noise'), quality=95, progressive=True, subsampling=0)):
        assert np.array_equal(model.visual.decode_jpeg(data).cpu().numpy(), _pil(data))
    with pytest.raises(_lib.OakeError):
        model.visual.decode_jpeg(_cmyk(_synth(32, 32, 'grad'), quality=80))
    with pytest.raises(_lib.OakeError):
        model.visual.decode_jpeg(b'not a jpeg at all')
