"""The three OAKE modes end to end on the GPU (product encoder through the C ABI) against the CPU
oracle: same synthetic COCO tree, same weights -> every .pth payload within fp16 tolerance."""
import ctypes as C
import pathlib

import numpy as np
import pytest
import torch

from oadp_amd import _lib, clip
from oadp_amd.config import Config
from oadp_amd.oake import blocks, globals as globals_, objects
from oadp_amd.weights import synthetic_state_dict

from . import _synth

pytestmark = pytest.mark.gpu
SIZES = [(300, 260), (224, 224), (500, 375), (250, 340), (100, 90), (640, 480)]


def _run_both(validator_cls, coco, tmp_path, tag, prep_model, **kw):
    outs = {}
    for who in ('gpu', 'ref'):
        out = tmp_path / f'{tag}_{who}'
        dl = Config(dataset=dict(root=coco['root'], annFile=coco['annFile'], output_dir=str(out),
                                 transform=_synth.preprocess(), **kw.get('dataset', {})), num_workers=0)
        if who == 'gpu':
            model, _ = clip.load(synthetic_state_dict(**_synth.TINY), max_batch=64)
            device = 'cuda:0'
        else:
            model, device = _synth.OracleModel(), 'cpu'
        prep_model(model, who)
        v = validator_cls(tag, model, dataloader=dl, device=device, **kw.get('validator', {}))
        v.run()
        outs[who] = out
    return outs


def _cmp(a, b):
    a, b = a.float(), b.float()
    assert a.shape == b.shape
    if a.numel():
        torch.testing.assert_close(a, b, rtol=1e-3, atol=1.5e-3)


def test_globals_blocks_objects_files_match_oracle(cuda, tmp_path):
    coco = _synth.make_coco(tmp_path / 'coco', SIZES)

    outs = _run_both(globals_.Validator, coco, tmp_path, 'globals', lambda m, w: None,
                     validator=dict(batch_size=4))
    for id_ in coco['ids']:
        g, r = (torch.load(outs[k] / f'{id_:012d}.pth', 'cpu') for k in ('gpu', 'ref'))
        assert g.dtype == torch.float16 and g.shape == (64,)
        _cmp(g, r)

    outs = _run_both(blocks.Validator, coco, tmp_path, 'blocks', lambda m, w: None,
                     validator=dict(batch_size=40))
    for id_ in coco['ids']:
        g, r = (torch.load(outs[k] / f'{id_:012d}.pth', 'cpu') for k in ('gpu', 'ref'))
        assert torch.equal(g['bboxes'], r['bboxes'])
        _cmp(g['embeddings'], r['embeddings'])

    def surgery(model, who):
        if who == 'ref':
            model.visual.objects_mode()
        else:
            v = model.visual
            v.positional_embedding = v.interpolate_positional_embedding((v.grid * 2,) * 2)
            v.grid *= 2
            v.conv1.stride = tuple(s // 2 for s in v.conv1.stride)
            v.conv1.padding = ((v.patch_size - 1) // 2,) * 2
            v.object_stream = True

    outs = _run_both(objects.Validator, coco, tmp_path, 'objects', surgery,
                     dataset=dict(type='COCODataset', proposal_file=coco['proposal_file'],
                                  proposal_sorted=True),
                     validator=dict(mini_batch_size=10, batch_size=30))
    for id_ in coco['ids']:
        g, r = (torch.load(outs[k] / f'{id_:012d}.pth', 'cpu') for k in ('gpu', 'ref'))
        assert torch.equal(g['bboxes'], r['bboxes']) and torch.equal(g['objectness'], r['objectness'])
        _cmp(g['embeddings'], r['embeddings'])


def test_device_preprocess_matches_host_preprocess(cuda, tmp_path, monkeypatch):
    """device_preprocess=True (uint8 upload, GPU crop / Pillow-exact resize / normalise) writes the
    same .pth payloads, bit for bit, as the PIL path of the reference's DataLoader workers."""
    monkeypatch.delenv('DRY_RUN', raising=False)
    coco = _synth.make_coco(tmp_path / 'coco', SIZES + [(1000, 700)])
    sd = synthetic_state_dict(**_synth.TINY)

    def run(cls, tag, dev_pre, surgery=False, dataset=None, validator=None):
        out = tmp_path / f'{tag}_{int(dev_pre)}'
        model, pre = clip.load(sd, max_batch=64)
        if surgery:
            v = model.visual
            v.positional_embedding = v.interpolate_positional_embedding((v.grid * 2,) * 2)
            v.grid *= 2
            v.conv1.stride = tuple(s // 2 for s in v.conv1.stride)
            v.conv1.padding = ((v.patch_size - 1) // 2,) * 2
            v.object_stream = True
        dl = Config(dataset=dict(root=coco['root'], annFile=coco['annFile'], output_dir=str(out),
                                 transform=pre, device_preprocess=dev_pre, **(dataset or {})), num_workers=0)
        cls(tag, model, dataloader=dl, device='cuda:0', **(validator or {})).run()
        return out

    for cls, tag, kw in [
            (globals_.Validator, 'g', {}),
            (blocks.Validator, 'b', dict(validator=dict(batch_size=64))),
            (objects.Validator, 'o', dict(surgery=True, dataset=dict(
                type='COCODataset', proposal_file=coco['proposal_file'], proposal_sorted=True),
                validator=dict(mini_batch_size=16, batch_size=32)))]:
        host, dev = run(cls, tag, False, **kw), run(cls, tag, True, **kw)
        for id_ in coco['ids']:
            a, b = torch.load(host / f'{id_:012d}.pth', 'cpu'), torch.load(dev / f'{id_:012d}.pth', 'cpu')
            if isinstance(a, dict):
                assert a.keys() == b.keys()
                for k in a:
                    assert torch.equal(a[k], b[k]), (tag, id_, k)
            else:
                assert torch.equal(a, b), (tag, id_)


def test_device_decode_matches_host_decode(cuda, tmp_path, monkeypatch):
    """device_decode=True (workers read file bytes; baseline JPEGs decoded by csrc/jpeg.hip, the rest by
    PIL) writes the same .pth payloads, bit for bit, as PIL decode + host preprocessing."""
    monkeypatch.delenv('DRY_RUN', raising=False)
    coco = _synth.make_coco(tmp_path / 'coco', SIZES + [(1000, 700)], fmt='jpg')
    sd = synthetic_state_dict(**_synth.TINY)

    def run(cls, tag, dev, dataset=None, validator=None):
        out = tmp_path / f'{tag}_{int(dev)}'
        model, pre = clip.load(sd, max_batch=64)
        dl = Config(dataset=dict(root=coco['root'], annFile=coco['annFile'], output_dir=str(out),
                                 transform=pre, device_decode=dev, **(dataset or {})), num_workers=0)
        cls(tag, model, dataloader=dl, device='cuda:0', **(validator or {})).run()
        return out

    for cls, tag, kw in [(globals_.Validator, 'g', {}),
                         (blocks.Validator, 'b', dict(validator=dict(batch_size=64)))]:
        host, dev = run(cls, tag, False, **kw), run(cls, tag, True, **kw)
        for id_ in coco['ids']:
            a, b = torch.load(host / f'{id_:012d}.pth', 'cpu'), torch.load(dev / f'{id_:012d}.pth', 'cpu')
            if isinstance(a, dict):
                for k in a:
                    assert torch.equal(a[k], b[k]), (tag, id_, k)
            else:
                assert torch.equal(a, b), (tag, id_)
    # 'strict' refuses the CMYK file instead of handing it to PIL
    with pytest.raises(ValueError):
        model, pre = clip.load(sd, max_batch=64)
        dl = Config(dataset=dict(root=coco['root'], annFile=coco['annFile'], output_dir=str(tmp_path / 's'),
                                 transform=pre, device_decode='strict'), num_workers=0)
        globals_.Validator('s', model, dataloader=dl, device='cuda:0').run()


def test_crop_normalize_kernel_is_bit_exact(cuda, lib):
    """GPU crop + ToTensor + Normalize of level-0 blocks == the host transform, bit for bit."""
    from oadp_amd.clip.preprocess import CLIP_MEAN, CLIP_STD
    from oracle import crops_ref
    import PIL.Image
    rng = np.random.default_rng(5)
    h, w = 375, 500
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    boxes = [(x, y, x + 224, y + 224) for x in crops_ref.partition(w) for y in crops_ref.partition(h)]
    boxes += [(-10, -20, 214, 204), (400, 300, 624, 524)]  # PIL zero-fill outside the image
    model, _ = clip.load(synthetic_state_dict(**_synth.TINY), max_batch=4)
    model.encode_image(torch.zeros(1, 3, 224, 224, device=cuda))  # creates the handle
    handle = model.visual._handle
    d_img = torch.from_numpy(img).to(cuda)
    d_boxes = torch.tensor(boxes, dtype=torch.int32, device=cuda)
    out = torch.empty(len(boxes), 3, 224, 224, device=cuda)
    mean = (C.c_float * 3)(*CLIP_MEAN)
    std = (C.c_float * 3)(*CLIP_STD)
    rc = lib.oake_crop_normalize(handle, d_img.data_ptr(), h, w, d_boxes.data_ptr(), len(boxes), 224,
                                 mean, std, out.data_ptr(), _lib.OAKE_F32,
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    pil = PIL.Image.fromarray(img)
    pre = _synth.preprocess()
    ref = torch.stack([pre(pil.crop(b)) for b in boxes])
    assert torch.equal(out.cpu(), ref)
