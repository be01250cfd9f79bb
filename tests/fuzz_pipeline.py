"""Random image trees through the three validators (GPU box; hand-run): 4..10 images of random size (1..900 px,
some extreme aspect ratios, some below one block), PNG or JPEG flavours, random proposals (some images with none that
survive the 4-px filter), random batch_size / mini_batch_size / lanes — the .pth payloads written with
device_preprocess / device_decode must equal, bit for bit, those written from the host (PIL) front end, and the
file set must be complete.  usage: python tests/fuzz_pipeline.py [n_trees=12] [seed=0]"""
import os, pathlib, pickle, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oadp_amd import clip
from oadp_amd.config import Config
from oadp_amd.oake import blocks, globals as globals_, objects
from oadp_amd.weights import synthetic_state_dict
from tests import _synth

n_trees = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
os.environ.pop('DRY_RUN', None)
sd = synthetic_state_dict(**_synth.TINY)
bad = files = 0

def surgery(model):
    v = model.visual
    v.positional_embedding = v.interpolate_positional_embedding((v.grid * 2,) * 2)
    v.grid *= 2
    v.conv1.stride = tuple(s // 2 for s in v.conv1.stride)
    v.conv1.padding = ((v.patch_size - 1) // 2,) * 2
    v.object_stream = True

for t in range(n_trees):
    root = pathlib.Path(tempfile.mkdtemp(prefix='oake_fuzz_'))
    try:
        sizes = []
        for _ in range(int(rng.integers(4, 11))):
            r = rng.random()
            if r < 0.15:
                sizes.append((int(rng.integers(1, 10)), int(rng.integers(1, 900))))
            elif r < 0.3:
                sizes.append((int(rng.integers(1, 900)), int(rng.integers(1, 10))))
            else:
                sizes.append((int(rng.integers(1, 900)), int(rng.integers(1, 700))))
        fmt = 'jpg' if rng.random() < 0.5 else 'png'
        coco = _synth.make_coco(root / 'coco', sizes, proposals_per_image=int(rng.integers(1, 30)), seed=int(rng.integers(1 << 30)), fmt=fmt)
        props = pickle.load(open(coco['proposal_file'], 'rb'))
        for i in range(len(props)):
            if rng.random() < 0.2:
                props[i][:, 2] = props[i][:, 0] + 2.0  # nothing survives min_wh=(4, 4)
        pickle.dump(props, open(coco['proposal_file'], 'wb'))
        bs = int(rng.choice([1, 3, 16, 64, 300]))
        mb = int(rng.choice([1, 7, 32]))
        streams = int(rng.choice([1, 2, 3]))
        modes = [('host', {}), ('dev_pre', dict(device_preprocess=True))]
        if fmt == 'jpg':
            modes.append(('dev_dec', dict(device_decode=True)))
        for cls, tag, extra_ds, extra_v, surg in (
                (globals_.Validator, 'globals', {}, {}, False),
                (blocks.Validator, 'blocks', {}, {}, False),
                (objects.Validator, 'objects', dict(type='COCODataset', proposal_file=coco['proposal_file'], proposal_sorted=True),
                 dict(mini_batch_size=mb), True)):
            outs = {}
            for name, kw in modes:
                out = root / f'{tag}_{name}'
                model, pre = clip.load(sd, max_batch=64)
                if surg:
                    surgery(model)
                dl = Config(dataset=dict(root=coco['root'], annFile=coco['annFile'], output_dir=str(out), transform=pre,
                                         **extra_ds, **kw), num_workers=0)
                v = cls(tag, model, dataloader=dl, device='cuda:0', batch_size=bs, streams=streams, **extra_v)
                v.run()
                outs[name] = out
            for id_ in coco['ids']:
                ref = torch.load(outs['host'] / f'{id_:012d}.pth', 'cpu')
                for name in outs:
                    if name == 'host':
                        continue
                    p = outs[name] / f'{id_:012d}.pth'
                    files += 1
                    if not p.exists():
                        bad += 1; print('MISSING', tag, name, id_, sizes); continue
                    got = torch.load(p, 'cpu')
                    same = (all(torch.equal(ref[k], got[k]) for k in ref) and ref.keys() == got.keys()) if isinstance(ref, dict) else torch.equal(ref, got)
                    if not same:
                        bad += 1
                        w, h = next(s for s, i in zip(sizes, sorted(coco['ids'])) if True)
                        print('MISMATCH', tag, name, 'id', id_, 'fmt', fmt, 'bs', bs, 'mb', mb, 'streams', streams, 'sizes', sizes)
    except Exception as e:
        bad += 1
        import traceback
        print('RAISED', repr(e)[:300], 'sizes', sizes, 'fmt', fmt); traceback.print_exc(limit=4)
    finally:
        shutil.rmtree(root, ignore_errors=True)
print(f'fuzz_pipeline seed {seed}: {n_trees} trees, {files} device-front-end files compared with the host front end, {bad} failures')
sys.exit(1 if bad else 0)
