"""Consumer side: the packed, memory-mapped access layer returns exactly what the
per-image .pth files hold, and LoadCLIPFeatures (oadp/dp/datasets.py:137-214) gives the same results
through either."""
import json
import pathlib
import pickle

import numpy as np
import pytest
import torch

from oadp_amd.dp import LoadCLIPFeatures, PackAccessLayer, PthAccessLayer, pack
from oadp_amd.dp.features import pairwise_intersection
from oadp_amd.oake.base import atomic_save


@pytest.fixture()
def oake_root(tmp_path):
    g = torch.Generator().manual_seed(0)
    ids = [9, 25, 139, 285, 632]
    for mode in ('globals', 'blocks', 'objects'):
        (tmp_path / mode / 'train2017').mkdir(parents=True)
    for i, id_ in enumerate(ids):
        key = f'{id_:012d}'
        atomic_save(torch.randn(1, 512, generator=g).half(), tmp_path / 'globals' / 'train2017' / f'{key}.pth')
        nb = 3 + 2 * i
        xy = torch.rand(nb, 2, generator=g) * 300
        atomic_save(dict(embeddings=torch.randn(nb, 512, generator=g).half(),
                         bboxes=torch.cat([xy, xy + 224], 1).half()),
                    tmp_path / 'blocks' / 'train2017' / f'{key}.pth')
        no = 0 if i == 2 else 4 + i   # an image without proposals: empty tensors
        xy = torch.rand(no, 2, generator=g) * 400
        wh = torch.rand(no, 2, generator=g) * 60   # some boxes narrower than 4 px
        atomic_save(dict(embeddings=torch.randn(no, 512, generator=g).half(),
                         bboxes=torch.cat([xy, xy + wh], 1).half(),
                         objectness=torch.rand(no, 1, generator=g).half()),
                    tmp_path / 'objects' / 'train2017' / f'{key}.pth')
    return tmp_path, ids


def _same(a, b):
    if isinstance(a, dict):
        assert list(a) == list(b)
        for k in a:
            _same(a[k], b[k])
    else:
        assert a.dtype == b.dtype and a.shape == b.shape
        assert torch.equal(a, b)


def test_pack_matches_pth_files(oake_root):
    root, ids = oake_root
    for mode in ('globals', 'blocks', 'objects'):
        blob = pack(str(root / mode), 'train2017')
        pth, pk = PthAccessLayer(str(root / mode), 'train2017'), PackAccessLayer(str(root / mode), 'train2017')
        assert list(pth) == list(pk) == [f'{i:012d}' for i in ids] and len(pk) == len(ids)
        for key in pth:
            _same(pth[key], pk[key])
        index = json.loads((blob.parent / (blob.name + '.json')).read_text())['index']
        assert all(off % 64 == 0 for e in index.values() for *_, off in e)
        with pytest.raises(KeyError):
            pk['000000000000']
        with pytest.raises(KeyError):
            pth['000000000000']


def test_pack_views_are_zero_copy_and_picklable(oake_root):
    root, ids = oake_root
    pack(str(root / 'blocks'), 'train2017')
    pk = PackAccessLayer(str(root / 'blocks'), 'train2017')
    key = f'{ids[1]:012d}'
    a, b = pk[key]['embeddings'], pk[key]['embeddings']
    assert a.data_ptr() == b.data_ptr()                      # views of the one map
    clone = pickle.loads(pickle.dumps(pk))                   # into a dataloader worker
    _same(clone[key], pk[key])
    private = PackAccessLayer(str(root / 'blocks'), 'train2017', copy=True)[key]['embeddings']
    private += 1                                             # writable, and the pack is untouched
    _same(PackAccessLayer(str(root / 'blocks'), 'train2017')[key], PthAccessLayer(str(root / 'blocks'), 'train2017')[key])


def test_truncated_pack_is_rejected(oake_root):
    root, _ = oake_root
    blob = pack(str(root / 'globals'), 'train2017')
    blob.write_bytes(blob.read_bytes()[:-8])
    with pytest.raises(ValueError, match='truncated'):
        PackAccessLayer(str(root / 'globals'), 'train2017')


def test_pack_rejects_foreign_payloads(tmp_path):
    (tmp_path / 'x').mkdir()
    torch.save(dict(a=[1, 2, 3]), tmp_path / 'x' / 'k.pth')
    with pytest.raises(TypeError):
        pack(str(tmp_path), 'x')


def test_pairwise_intersection_against_loops():
    g = torch.Generator().manual_seed(1)
    a = torch.rand(7, 2, generator=g) * 50
    a = torch.cat([a, a + torch.rand(7, 2, generator=g) * 40], 1)
    b = torch.rand(5, 2, generator=g) * 50
    b = torch.cat([b, b + torch.rand(5, 2, generator=g) * 40], 1)
    got = pairwise_intersection(a, b)
    for i in range(7):
        for j in range(5):
            w = min(a[i, 2], b[j, 2]) - max(a[i, 0], b[j, 0])
            h = min(a[i, 3], b[j, 3]) - max(a[i, 1], b[j, 1])
            want = max(w, 0) * max(h, 0)
            assert abs(got[i, j] - want) < 1e-4


def _sample(id_):
    return dict(img_info=dict(id=id_), bbox_fields=['gt_bboxes'],
                gt_bboxes=np.array([[10, 10, 120, 90], [300, 280, 420, 400], [0, 0, 5, 5]], np.float32),
                gt_labels=np.array([3, 64, 70]))   # 70 >= num_all: a pseudo label


@pytest.mark.parametrize('layer', ['PthAccessLayer', 'PackAccessLayer'])
def test_load_clip_features(oake_root, layer):
    root, ids = oake_root
    for mode in ('globals', 'blocks', 'objects'):
        pack(str(root / mode), 'train2017')
    step = LoadCLIPFeatures(default=dict(task_name='train2017', type=layer),
                            globals_=dict(data_root=str(root / 'globals')),
                            blocks=dict(data_root=str(root / 'blocks')),
                            objects=dict(data_root=str(root / 'objects')))
    for id_ in ids:
        key = f'{id_:012d}'
        out = step(_sample(id_))
        g = torch.load(root / 'globals' / 'train2017' / f'{key}.pth')
        b = torch.load(root / 'blocks' / 'train2017' / f'{key}.pth')
        o = torch.load(root / 'objects' / 'train2017' / f'{key}.pth')
        assert out['bbox_fields'] == ['gt_bboxes', 'block_bboxes', 'object_bboxes']
        assert out['clip_global'].shape == (512,) and torch.equal(out['clip_global'], g[0])
        assert torch.equal(out['clip_blocks'], b['embeddings'])
        assert out['block_bboxes'].dtype == np.float32
        assert np.array_equal(out['block_bboxes'], b['bboxes'].float().numpy())
        labels = np.zeros((b['bboxes'].shape[0], 65), bool)   # brute force, pseudo label 70 ignored
        for bi, bb in enumerate(b['bboxes'].float().tolist()):
            for gt, lab in zip(_sample(id_)['gt_bboxes'][:2].tolist(), (3, 64)):
                if min(bb[2], gt[2]) > max(bb[0], gt[0]) and min(bb[3], gt[3]) > max(bb[1], gt[1]):
                    labels[bi, lab] = True
        assert out['block_labels'].dtype == bool and np.array_equal(out['block_labels'], labels)
        wh = o['bboxes'][:, 2:] - o['bboxes'][:, :2]
        keep = (wh[:, 0] >= 4) & (wh[:, 1] >= 4)
        assert torch.equal(out['clip_objects'], o['embeddings'][keep])
        assert np.array_equal(out['object_bboxes'], o['bboxes'][keep].float().numpy())


def test_load_clip_features_without_annotations_and_dry_run(oake_root, monkeypatch):
    root, ids = oake_root
    step = LoadCLIPFeatures(default=dict(task_name='train2017', type='PthAccessLayer'),
                            blocks=dict(data_root=str(root / 'blocks')))
    out = step(dict(img_info=dict(id=ids[0]), bbox_fields=[]))
    assert 'block_labels' not in out and 'clip_global' not in out and out['bbox_fields'] == ['block_bboxes']
    with pytest.raises(ValueError):
        LoadCLIPFeatures(default=dict(task_name='train2017', type='PthAccessLayer'))
    monkeypatch.setenv('DRY_RUN', '1')
    step = LoadCLIPFeatures(default=dict(task_name='train2017', type='PthAccessLayer'),
                            globals_=dict(data_root=str(root / 'globals')),
                            objects=dict(data_root=str(root / 'objects')))
    out = step(dict(img_info=dict(id=123456), bbox_fields=[]))   # DRY_RUN: one common key for every sample
    assert torch.equal(out['clip_global'], torch.load(root / 'globals' / 'train2017' / f'{ids[0]:012d}.pth')[0])


def test_val_split_switch(oake_root, monkeypatch):
    root, ids = oake_root
    (root / 'globals' / 'val2017').mkdir()
    atomic_save(torch.ones(1, 512).half(), root / 'globals' / 'val2017' / f'{ids[0]:012d}.pth')
    monkeypatch.setenv('TRAIN_WITH_VAL_DATASET', '1')
    step = LoadCLIPFeatures(default=dict(task_name='train2017', type='PthAccessLayer'),
                            globals_=dict(data_root=str(root / 'globals')))
    assert torch.equal(step(dict(img_info=dict(id=ids[0]), bbox_fields=[]))['clip_global'], torch.ones(512).half())


# ---- the validators' direct pack writer (second half) -------------------
def _validators(coco, root, writer, shard=None, monkeypatch=None):
    from oadp_amd.config import Config
    from oadp_amd.oake import blocks, globals as globals_, objects
    from . import _synth
    if monkeypatch is not None:
        if shard:
            monkeypatch.setenv('OAKE_SHARD', shard)
        else:
            monkeypatch.delenv('OAKE_SHARD', raising=False)
        monkeypatch.setenv('OAKE_CPU_AFFINITY', '0')

    def dl(mode, **extra):
        return Config(dataset=dict(root=coco['root'], annFile=coco['annFile'], output_dir=str(root / mode / 'train2017'),
                                   transform=_synth.preprocess(), **extra), num_workers=0)
    counts = {}
    counts['globals'] = globals_.Validator('g', _synth.OracleModel(), dataloader=dl('globals'), batch_size=4,
                                           device='cpu', writer=writer).run().images
    counts['blocks'] = blocks.Validator('b', _synth.OracleModel(), dataloader=dl('blocks'), batch_size=8,
                                        device='cpu', writer=writer).run().images
    model = _synth.OracleModel()
    model.visual.objects_mode()
    counts['objects'] = objects.Validator(
        'o', model, dataloader=dl('objects', type='COCODataset', proposal_file=coco['proposal_file'], proposal_sorted=True),
        mini_batch_size=7, batch_size=16, device='cpu', writer=writer).run().images
    return counts


def test_validators_write_packs_directly(tmp_path, monkeypatch):
    """writer='pack': the three validators append every flush to one blob + index per tree — no per-image files —
    and PackAccessLayer returns, key for key and bit for bit, what PthAccessLayer returns for the default writer."""
    from . import _synth
    monkeypatch.delenv('DRY_RUN', raising=False)
    coco = _synth.make_coco(tmp_path / 'coco', [(300, 260), (224, 224), (500, 375), (250, 340), (100, 90), (640, 480)],
                            proposals_per_image=15)
    pth_root, pack_root = tmp_path / 'pth', tmp_path / 'pack'
    n = len(coco['ids'])
    assert _validators(coco, pth_root, 'pth', monkeypatch=monkeypatch) == dict(globals=n, blocks=n, objects=n)
    assert _validators(coco, pack_root, 'pack', monkeypatch=monkeypatch) == dict(globals=n, blocks=n, objects=n)
    for mode in ('globals', 'blocks', 'objects'):
        assert not list((pack_root / mode / 'train2017').glob('*.pth'))            # no small files
        assert (pack_root / mode / 'train2017.pack').exists() and (pack_root / mode / 'train2017.pack.json').exists()
        a, b = PthAccessLayer(str(pth_root / mode), 'train2017'), PackAccessLayer(str(pack_root / mode), 'train2017')
        assert sorted(a) == sorted(b) == [f'{i:012d}' for i in coco['ids']]
        for key in a:
            _same(a[key], b[key])
    # resume: a second run finds every key in the blob and encodes nothing; the blob is unchanged
    before = (pack_root / 'blocks' / 'train2017.pack').read_bytes()
    assert _validators(coco, pack_root, 'pack', monkeypatch=monkeypatch) == dict(globals=0, blocks=0, objects=0)
    assert (pack_root / 'blocks' / 'train2017.pack').read_bytes() == before
    # ... and LoadCLIPFeatures over the direct packs == over the .pth trees
    cfg = lambda root: dict(globals_=dict(data_root=str(root / 'globals')), blocks=dict(data_root=str(root / 'blocks')),
                            objects=dict(data_root=str(root / 'objects')))
    s_pth = LoadCLIPFeatures(default=dict(task_name='train2017', type='PthAccessLayer'), **cfg(pth_root))
    s_pack = LoadCLIPFeatures(default=dict(task_name='train2017', type='PackAccessLayer'), **cfg(pack_root))
    for id_ in coco['ids']:
        x, y = s_pth(_sample(id_)), s_pack(_sample(id_))
        assert list(x) == list(y)
        for k in x:
            if isinstance(x[k], torch.Tensor):
                assert torch.equal(x[k], y[k])
            elif isinstance(x[k], np.ndarray):
                assert np.array_equal(x[k], y[k])


def test_sharded_packs_and_a_killed_writer(tmp_path, monkeypatch):
    """Two shards (OAKE_SHARD=r/2) write train2017.r0of2.pack / .r1of2.pack; the access layer reads their union.  A
    writer killed between index checkpoints leaves a tail behind the index: ignored by readers, truncated on resume."""
    from oadp_amd.packfile import PackWriter, blob_path
    from . import _synth
    monkeypatch.delenv('DRY_RUN', raising=False)
    coco = _synth.make_coco(tmp_path / 'coco', [(300, 260), (224, 224), (250, 340), (100, 90), (320, 240)])
    root = tmp_path / 'oake'
    total = 0
    for r in range(2):
        total += _validators(coco, root, 'pack', shard=f'{r}/2', monkeypatch=monkeypatch)['globals']
    assert total in (len(coco['ids']), len(coco['ids']) + 1)  # (the sampler pads 5 images to 6 by wrap-around)
    assert sorted(p.name for p in (root / 'globals').glob('*.pack')) == ['train2017.r0of2.pack', 'train2017.r1of2.pack']
    layer = PackAccessLayer(str(root / 'globals'), 'train2017')
    assert sorted(layer) == [f'{i:012d}' for i in coco['ids']]
    monkeypatch.delenv('OAKE_SHARD')
    blob = blob_path(tmp_path / 'k' / 'train2017')
    w = PackWriter(blob, checkpoint=2)
    for i in range(5):  # index checkpoints after keys 2 and 4; key 5 stays behind the index
        w.submit(torch.full((3,), float(i)).half(), pathlib.Path(f'{i:012d}.pth'))
    w._f.flush()  # (killed here: no close)
    r = PackAccessLayer(str(tmp_path / 'k'), 'train2017')
    assert sorted(r) == [f'{i:012d}' for i in range(4)] and torch.equal(r['000000000003'], torch.full((3,), 3.0).half())
    w2 = PackWriter(blob)  # resume: the orphaned tail goes, the four indexed keys stay
    assert w2.keys == {f'{i:012d}' for i in range(4)} and blob.stat().st_size == w2._offset
    w2.submit(torch.full((3,), 9.0).half(), pathlib.Path('000000000009.pth'))
    w2.submit(torch.full((3,), 1.0).half(), pathlib.Path('000000000001.pth'))  # a duplicate key is dropped
    w2.close()
    r = PackAccessLayer(str(tmp_path / 'k'), 'train2017')
    assert len(r) == 5 and torch.equal(r['000000000009'], torch.full((3,), 9.0).half())
    assert torch.equal(r['000000000001'], torch.full((3,), 1.0).half())


def test_pack_layer_refuses_shards_of_two_sweeps(tmp_path):
    """Advisor r04: shards left over from a sweep with another world size (or a stale single blob beside fresh
    shards) must not silently shadow fresh features; an incomplete shard set warns."""
    import warnings
    from oadp_amd.packfile import PackWriter, blob_path
    out = tmp_path / 'globals' / 'train2017'
    out.parent.mkdir(parents=True)

    def write(rank, world, keys):
        w = PackWriter(blob_path(out, rank, world))
        for k in keys:
            w.submit(torch.full((4,), float(k)).half(), out / f'{k:012d}.pth')
        w.close()

    write(0, 2, [0, 2])
    write(1, 2, [1, 3])
    layer = PackAccessLayer(str(tmp_path / 'globals'), 'train2017')
    assert sorted(layer) == [f'{k:012d}' for k in range(4)] and layer['000000000003'][0].item() == 3
    write(0, 4, [0])  # a relaunch with another world size
    with pytest.raises(ValueError, match='more than one sweep'):
        PackAccessLayer(str(tmp_path / 'globals'), 'train2017')
    for f in (tmp_path / 'globals').glob('train2017.r0of4.pack*'):
        f.unlink()
    write(0, 1, [7])  # a stale single blob beside the shards
    with pytest.raises(ValueError, match='more than one sweep'):
        PackAccessLayer(str(tmp_path / 'globals'), 'train2017')
    for f in (tmp_path / 'globals').glob('train2017.pack*'):
        f.unlink()
    for f in (tmp_path / 'globals').glob('train2017.r1of2.pack*'):
        f.unlink()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        layer = PackAccessLayer(str(tmp_path / 'globals'), 'train2017')
    assert sorted(layer) == ['000000000000', '000000000002'] and any('missing ranks' in str(x.message) for x in w)
