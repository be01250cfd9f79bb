"""Ragged / empty / extreme inputs through the device paths (the edge cases of the crop math
and of the encoder boundary), each against the host (PIL / oracle) twin of the same call.

  * images far from the 224-px working size: 1x1, 1xN, Nx1, thin strips, > 2048 px — the whole-image
    preprocess and the blocks pyramid must stay Pillow-exact where a resample support is wider than the source;
  * images smaller than one block (no tiles: block 0 only) and exactly one block;
  * an objects-mode image whose proposals are all filtered out, alone in a flush and between others;
  * zero-crop calls into the encoder boundary.
"""
import pickle

import numpy as np
import PIL.Image
import pytest
import torch

from oadp_amd import clip
from oadp_amd.clip.preprocess import Preprocess
from oadp_amd.config import Config
from oadp_amd.oake import blocks, objects
from oadp_amd.weights import synthetic_state_dict

from . import _synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def tiny_model(cuda):
    model, pre = clip.load(synthetic_state_dict(**_synth.TINY), max_batch=64)
    return model, pre


def _img(w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)


EXTREME = [(1, 1), (1, 300), (300, 1), (2, 2000), (2500, 3), (7, 5), (223, 225), (224, 224), (2600, 1900)]


@pytest.mark.parametrize('squash', [False, True])
def test_whole_image_preprocess_extreme_sizes(tiny_model, cuda, squash):
    """preprocess(image) on the device == the PIL transform, bit for bit, for sources from 1x1 to 2600x1900."""
    model, _ = tiny_model
    host = Preprocess(224, squash=squash)
    arrs = [_img(w, h, 11 + i) for i, (w, h) in enumerate(EXTREME)]
    dev = [torch.from_numpy(a).to(cuda) for a in arrs]
    got = model.visual.crop_resize_normalize_batch(dev, [[(0, 0, a.shape[1], a.shape[0])] for a in arrs],
                                                   squash=squash, out_dtype=torch.float32)
    assert got.shape == (len(arrs), 3, 224, 224)
    for i, a in enumerate(arrs):
        ref = host(PIL.Image.fromarray(a))
        assert torch.equal(got[i].cpu(), ref), EXTREME[i]


def test_blocks_of_images_smaller_than_a_block(tiny_model, cuda):
    """No pyramid level holds a 224-px tile: block 0 (the whole image) is the only crop; 224x224 and
    225x224 are the first sizes with tiles.  Counts and pixels against the host dataset."""
    model, pre = tiny_model
    ds = blocks.Dataset.__new__(blocks.Dataset)
    ds._r, ds._s, ds._rescale = 224, 112, 1.5
    ds.transform = pre
    # (3, 700) and (5, 1200): more than 100x taller than wide — Pillow resamples those vertically first
    sizes = [(10, 10), (223, 500), (500, 223), (224, 224), (225, 224), (1, 1), (3, 700), (5, 1200)]
    arrs = [_img(w, h, 40 + i) for i, (w, h) in enumerate(sizes)]
    out, counts = model.visual.blocks_batch([torch.from_numpy(a).to(cuda) for a in arrs], block_size=224,
                                            max_stride=112, rescale=1.5, out_dtype=torch.float32)
    import pathlib
    exp_counts, i0 = [], 0
    for a, (w, h) in zip(arrs, sizes):
        host = ds._preprocess(0, pathlib.Path('x'), PIL.Image.fromarray(a))
        k = host.blocks.shape[0]
        exp_counts.append(k)
        assert torch.equal(out[i0:i0 + k].cpu(), host.blocks), (w, h)
        i0 += k
    assert counts == exp_counts == [1, 1, 1, 2, 3, 1, 1, 1]


def _objects_run(tmp_path, tag, props, sizes, device_preprocess, model, pre):
    coco = _synth.make_coco(tmp_path / f'coco_{tag}', sizes)
    with open(coco['proposal_file'], 'wb') as f:
        pickle.dump(props, f)
    out = tmp_path / f'out_{tag}_{int(device_preprocess)}'
    dl = Config(dataset=dict(type='COCODataset', root=coco['root'], annFile=coco['annFile'], output_dir=str(out),
                             transform=pre, proposal_file=coco['proposal_file'], proposal_sorted=True,
                             device_preprocess=device_preprocess), num_workers=0)
    v = objects.Validator('objects', model, dataloader=dl, device='cuda:0', mini_batch_size=8, batch_size=16)
    v.run()
    return coco, out, v


def test_objects_image_whose_proposals_are_all_filtered(cuda, tmp_path, monkeypatch):
    """indices(min_wh=(4, 4)) can drop every proposal of an image (reference objects.py:160-164): the file is
    written with empty tensors — alone in its flush and between images that do have crops, host and device
    preprocessing alike."""
    monkeypatch.delenv('DRY_RUN', raising=False)
    model, pre = clip.load(synthetic_state_dict(**_synth.TINY), max_batch=64)
    v = model.visual
    v.positional_embedding = v.interpolate_positional_embedding((v.grid * 2,) * 2)
    v.grid *= 2
    v.conv1.stride = tuple(s // 2 for s in v.conv1.stride)
    v.conv1.padding = ((v.patch_size - 1) // 2,) * 2
    v.object_stream = True
    sizes = [(320, 240), (200, 150), (300, 300)]
    good = np.array([[10, 10, 100, 90, 0.9], [50, 40, 300, 200, 0.8], [0, 0, 20, 30, 0.7]], np.float32)
    tiny = np.array([[10, 10, 12, 90, 0.9], [50, 40, 300, 43, 0.8]], np.float32)   # w or h < 4
    none = np.zeros((0, 5), np.float32)
    for tag, props in (('mixed', [good, tiny, good]), ('all_empty', [tiny, none, tiny])):
        results = {}
        for dev_pre in (False, True):
            coco, out, val = _objects_run(tmp_path, tag, props, sizes, dev_pre, model, pre)
            results[dev_pre] = [torch.load(out / f'{id_:012d}.pth', 'cpu') for id_ in coco['ids']]
            assert val.counters.images == 3
            assert val.counters.crops == sum(3 for p in props if p is good)
        for a, b, p in zip(results[False], results[True], props):
            n = 3 if p is good else 0
            for r in (a, b):
                assert r['embeddings'].shape == (n, 64) and r['embeddings'].dtype == torch.float16
                assert r['bboxes'].shape == (n, 4) and r['objectness'].shape[0] == n
            assert torch.equal(a['embeddings'], b['embeddings']) and torch.equal(a['bboxes'], b['bboxes'])


def test_pyramid_resize_of_a_very_narrow_image(tiny_model, cuda):
    """oake_resize_u8 (the blocks pyramid step) on sources Pillow resamples vertically first, and on their
    wide counterparts, against PIL itself."""
    model, _ = tiny_model
    for w, h, ow, oh in [(3, 700, 2, 467), (5, 1200, 3, 800), (700, 3, 467, 2), (6, 601, 4, 401), (6, 600, 4, 400)]:
        a = _img(w, h, w + h)
        got = model.visual.resize_u8(torch.from_numpy(a).to(cuda), (ow, oh))
        ref = np.asarray(PIL.Image.fromarray(a).resize((ow, oh), PIL.Image.BICUBIC))
        assert np.array_equal(got.cpu().numpy(), ref), (w, h, ow, oh)


def test_zero_crop_calls_at_the_encoder_boundary(tiny_model, cuda):
    model, _ = tiny_model
    e = model.encode_image(torch.zeros(0, 3, 224, 224, device=cuda), normalize=True, out_dtype=torch.float16)
    assert e.shape == (0, 64) and e.dtype == torch.float16
    c = model.visual.crop_resize_normalize_batch([torch.zeros(30, 40, 3, dtype=torch.uint8, device=cuda)], [[]],
                                                 out_dtype=torch.float16)
    assert c.shape == (0, 3, 224, 224)
    c = model.visual.crop_resize_normalize_batch([], [], out_dtype=torch.float16)
    assert c.shape == (0, 3, 224, 224)
