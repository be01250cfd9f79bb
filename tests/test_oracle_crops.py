"""Pin the crop-math oracle (oracle/crops_ref.py) and the product's host code (oadp_amd.oake)
against the golden fixtures minted from the REFERENCE's own functions (tools/gen_golden.py).
Integer / index work is compared bit-exactly."""
import json
import pathlib
import types

import numpy as np
import PIL.Image
import pytest
import torch

from oadp_amd.clip.preprocess import Preprocess
from oadp_amd.oake import blocks as pblocks
from oadp_amd.oake import objects as pobjects
from oracle import crops_ref

GOLD = pathlib.Path(__file__).parent / 'golden'


@pytest.fixture(scope='module')
def gb():
    return json.loads((GOLD / 'blocks_partition.json').read_text())


@pytest.fixture(scope='module')
def go():
    return json.loads((GOLD / 'objects_masks_expand.json').read_text())


def _blocks_ds():
    ds = pblocks.Dataset.__new__(pblocks.Dataset)
    ds._r, ds._s, ds._rescale = 224, 112, 1.5
    ds.transform = Preprocess(224, squash=False)
    return ds


def _objects_ds():
    ds = pobjects.COCODataset.__new__(pobjects.COCODataset)
    ds._grid = 14
    ds._expand_mode = pobjects.ExpandMode.ADAPTIVE
    ds.transform = Preprocess(224, squash=False)
    return ds


def test_partition_all_lengths(gb):
    ds = _blocks_ds()
    assert len(gb['partition']) > 1500
    for k, expected in gb['partition'].items():
        assert crops_ref.partition(int(k)) == expected, k
        assert ds._partition(int(k)) == expected, k


def test_partition_known_answers():
    # SURVEY.md Appendix A.4 (reference outputs)
    known = {100: [], 223: [], 224: [0], 225: [0, 1], 336: [0, 112], 337: [0, 57, 113],
             480: [0, 86, 171, 256], 640: [0, 104, 208, 312, 416]}
    for n, exp in known.items():
        assert crops_ref.partition(n) == exp


def test_pyramid_tiles_and_bboxes(gb):
    ds = _blocks_ds()
    for img in gb['images']:
        w, h = img['size']
        tiles = crops_ref.partitions(w, h)
        assert [[a, b, c, d, e] for a, b, c, d, e in tiles] == img['tiles'], (w, h)
        bboxes = [list(crops_ref.block_bbox(sc, x, y)) for _, _, sc, x, y in tiles]
        assert bboxes == img['bboxes']
        allb = crops_ref.all_block_bboxes(w, h)
        assert np.array_equal(allb, np.asarray(img['batch_bboxes'], dtype=np.float32))
        assert allb.shape[0] == img['n_blocks']
        # product: same generator protocol as the reference
        got = [[im.size[0], im.size[1], sc, x, y] for im, sc, x, y in ds._partitions(PIL.Image.new('RGB', (w, h)))]
        assert got == img['tiles']
        assert [list(ds._bbox(sc, x, y)) for _, _, sc, x, y in got] == img['bboxes']


def test_five_level_image(gb):
    big = [i for i in gb['images'] if i['size'] == [1700, 1134]][0]
    assert big['n_blocks'] == 245 and len({t[2] for t in big['tiles']}) == 5


def test_blocks_preprocess_pixels():
    """Product Dataset._preprocess == reference _preprocess on a textured image (pixel level)."""
    z = np.load(GOLD / 'blocks_pixels.npz')
    ds = _blocks_ds()
    batch = ds._preprocess(0, pathlib.Path('x.pth'), PIL.Image.fromarray(z['image']))
    assert torch.equal(batch.bboxes, torch.from_numpy(z['bboxes']))
    assert torch.equal(batch.blocks, torch.from_numpy(z['blocks']))
    # oracle transform on block 0
    assert np.array_equal(crops_ref.preprocess_ref(PIL.Image.fromarray(z['image'])), z['blocks'][0])


def test_object_masks(go):
    ds = _objects_ds()
    assert len(go['masks']) > 80
    for case in go['masks']:
        fg, ob = tuple(case['foreground']), tuple(case['object'])
        exp = np.asarray(case['mask'], dtype=np.uint8)
        assert np.array_equal(crops_ref.object_mask(fg, ob), exp), case
        got = ds._mask(fg, ob)
        assert got.shape == (1, 1, 14, 14)
        assert np.array_equal(got.reshape(14, 14).numpy().astype(np.uint8), exp)


def test_mask_known_answer():
    # SURVEY.md Appendix A.5
    m = crops_ref.object_mask((10., 20., 60., 90.), (0, 0, 112, 112))
    assert m[:3].all() and m[12:].all()
    assert m[5].tolist() == [1, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1]


def test_expand_and_objects_preprocess(go, monkeypatch):
    monkeypatch.delenv('DRY_RUN', raising=False)
    ds = _objects_ds()
    for case in go['expand']:
        prop = np.asarray(case['proposals'], dtype=np.float32)
        keep = crops_ref.keep_min_wh(prop[:, :4])
        assert keep.tolist() == case['keep']
        exp = np.asarray(case['expanded'], dtype=np.float32)
        # float boxes: torch.sqrt (reference) and np.sqrt differ by 1 ulp on some inputs and
        # torch's vectorised sqrt is CPU-dependent, so floats are compared to 1e-4 px and the
        # integer crop boxes PIL derives from them (the actual crop indices) exactly.
        got = crops_ref.expand_adaptive(prop[keep, :4], case['image_size'])
        assert np.allclose(got, exp, rtol=0, atol=1e-3)
        pgot = ds._expand(torch.from_numpy(prop[keep, :4]), torch.tensor(case['image_size'])).numpy()
        assert np.allclose(pgot, exp, rtol=0, atol=1e-3)
        for a, b, c in zip(got, pgot, exp):
            assert crops_ref.pil_crop_box(a) == crops_ref.pil_crop_box(c) == crops_ref.pil_crop_box(b)
        # property: a box that fits lies inside the image; a larger one stays centred
        w, h = case['image_size']
        side = got[:, 2] - got[:, 0]
        fits = side <= min(w, h)
        assert (got[fits, 0] >= -1e-3).all() and (got[fits, 2] <= w + 1e-3).all()
        assert (got[fits, 1] >= -1e-3).all() and (got[fits, 3] <= h + 1e-3).all()


def test_objects_full_preprocess(go, monkeypatch):
    monkeypatch.delenv('DRY_RUN', raising=False)
    z = np.load(GOLD / 'objects_pixels.npz')
    case = [c for c in go['expand'] if c['image_size'] == [200, 150]][0]
    ds = _objects_ds()
    ds._proposals = {7: torch.from_numpy(z['proposals'])}
    batch = ds._preprocess(7, pathlib.Path('x.pth'), PIL.Image.fromarray(z['image']))
    assert batch.objects.shape[0] == case['n_objects']
    assert torch.equal(batch.objects[:6], torch.from_numpy(z['objects']))
    assert np.array_equal(batch.masks.reshape(-1, 14, 14).numpy().astype(np.uint8),
                          np.asarray(case['masks'], dtype=np.uint8))
    assert np.allclose(batch.bboxes.numpy(), np.asarray(case['bboxes'], dtype=np.float32))
    assert np.allclose(batch.objectness.numpy(), np.asarray(case['objectness'], dtype=np.float32))


def test_dry_run_keeps_five(monkeypatch, go):
    monkeypatch.setenv('DRY_RUN', 'True')
    z = np.load(GOLD / 'objects_pixels.npz')
    ds = _objects_ds()
    ds._proposals = {7: torch.from_numpy(z['proposals'])}
    batch = ds._preprocess(7, pathlib.Path('x.pth'), PIL.Image.fromarray(z['image']))
    assert batch.objects.shape[0] <= 5  # reference objects.py:166-167


def test_pil_crop_rounding():
    # banker's rounding, zero padding outside
    assert crops_ref.pil_crop_box((0.5, 1.5, 4.5, 6.5)) == (0, 2, 4, 6)
    img = PIL.Image.new('RGB', (4, 4), (9, 9, 9))
    c = np.asarray(img.crop((-2, -2, 2, 2)))
    assert c[0, 0].tolist() == [0, 0, 0] and c[3, 3].tolist() == [9, 9, 9]


def test_native_blocks_count_matches_reference_partitions(gb):
    """oake_blocks_count / oake_blocks_batch restate _partition and the pyramid walk in C++ (host side of
    the batched device path): integer-exact against the reference's own outputs and against the Python
    twin on random sizes."""
    import random
    from oadp_amd import _lib
    lib = _lib.load()
    for im in gb['images']:
        w, h = im['size']
        assert lib.oake_blocks_count(w, h, 224, 112, 1.5) == im['n_blocks'], (w, h)
    ds = pblocks.Dataset.__new__(pblocks.Dataset)
    rnd = random.Random(5)
    for r, s_, rescale in ((224, 112, 1.5), (224, 112, 1.25), (64, 48, 2.0)):
        ds._r, ds._s, ds._rescale = r, s_, rescale
        for _ in range(300):
            w, h = rnd.randint(1, 2600), rnd.randint(1, 2600)
            assert lib.oake_blocks_count(w, h, r, s_, rescale) == 1 + len(ds._level_tiles(w, h)), (w, h, r, s_, rescale)
    assert lib.oake_blocks_count(0, 10, 224, 112, 1.5) == -1 and lib.oake_blocks_count(10, 10, 224, 112, 1.0) == -1
