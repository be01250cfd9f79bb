"""oake/fastsave.py: template-patched torch.save archives for the per-image feature files."""
import io
import threading
import zipfile

import pytest
import torch

from oadp_amd.oake.base import atomic_save
from oadp_amd.oake.fastsave import FastSaver


def _ref(obj):
    b = io.BytesIO()
    torch.save(obj, b)
    return b.getvalue()


def _payloads(g):
    yield lambda: torch.randn(1, 512, generator=g).half()
    yield lambda: dict(embeddings=torch.randn(27, 512, generator=g).half(), bboxes=torch.rand(27, 4, generator=g).half())
    yield lambda: dict(embeddings=torch.randn(300, 512, generator=g).half(), bboxes=torch.rand(300, 4, generator=g).half(),
                       objectness=torch.rand(300, 1, generator=g).half())
    yield lambda: dict(embeddings=torch.zeros(0, 512).half(), bboxes=torch.zeros(0, 4).half(),
                       objectness=torch.zeros(0, 1).half())
    yield lambda: torch.randn(3, 5, generator=g).to(torch.bfloat16)
    yield lambda: dict(a=torch.randint(0, 9, (4, 4), generator=g), b=torch.rand(2, generator=g))


def _equal(a, b):
    if isinstance(a, dict):
        return list(a) == list(b) and all(_equal(a[k], b[k]) for k in a)
    return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b)


def test_archives_match_torch_save_up_to_the_serialization_id():
    g = torch.Generator().manual_seed(0)
    saver = FastSaver()
    for make in _payloads(g):
        for i in range(3):
            obj = make()
            data = saver.dumps(obj)
            assert data is not None
            assert _equal(torch.load(io.BytesIO(bytes(data))), obj)
            ref = _ref(obj)
            assert len(data) == len(ref)
            # only the informational serialization_id record (and its CRC copies) may differ
            zf = zipfile.ZipFile(io.BytesIO(ref))
            sid = next(x for x in zf.infolist() if x.filename.endswith('serialization_id'))
            differing = [k for k in range(len(ref)) if data[k] != ref[k]]
            assert all(k >= sid.header_offset for k in differing) and len(differing) <= 48
            for name in zf.namelist():   # every record passes the zip CRC check
                zipfile.ZipFile(io.BytesIO(bytes(data))).read(name)
    assert saver.hits > 0 and saver.misses == len(list(_payloads(g)))


def test_anything_unusual_goes_to_torch_save():
    saver = FastSaver()
    base = torch.arange(12.)
    for obj in ([1, 2, 3], 'text', dict(a=1), dict(a=torch.zeros(2), b=[1]), {},
                base[2:], base.view(3, 4).t(),                     # offset / non-contiguous views
                dict(a=base, b=base),                              # one storage, two tensors
                torch.zeros(2, requires_grad=True), {1: torch.zeros(2)}):
        assert saver.dumps(obj) is None
    sub = type('Sub', (torch.Tensor,), {})
    assert saver.dumps(torch.zeros(3).as_subclass(sub)) is None


def test_unverifiable_layout_disables_the_fast_path(monkeypatch):
    import oadp_amd.oake.fastsave as fs
    saver = FastSaver()
    monkeypatch.setattr(fs, '_same', lambda a, b: False)   # the read-back check fails
    t = torch.randn(1, 8)
    assert saver.dumps(t) is None and saver.dumps(torch.randn(1, 8)) is None
    assert saver.hits == 0


def test_atomic_save_uses_it_and_files_load(tmp_path):
    g = torch.Generator().manual_seed(1)
    for i in range(4):
        obj = dict(embeddings=torch.randn(5, 512, generator=g).half(), bboxes=torch.rand(5, 4, generator=g).half())
        atomic_save(obj, tmp_path / f'{i:012d}.pth')
        assert _equal(torch.load(tmp_path / f'{i:012d}.pth'), obj)
    atomic_save([1, 2], tmp_path / 'list.pth')                 # fallback path
    assert torch.load(tmp_path / 'list.pth') == [1, 2]
    assert not list(tmp_path.glob('*.tmp*'))


def test_concurrent_writers_share_templates():
    saver = FastSaver()
    errors = []

    def work(seed):
        g = torch.Generator().manual_seed(seed)
        try:
            for _ in range(200):
                t = torch.randn(1, 512, generator=g).half()
                d = saver.dumps(t)
                if not torch.equal(torch.load(io.BytesIO(bytes(d))), t):
                    errors.append('mismatch')
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=work, args=(s,)) for s in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors
