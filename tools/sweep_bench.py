"""End-to-end OAKE sweep throughput on a synthetic COCO-like JPEG set (GPU box): files -> decode ->
preprocess -> encoder -> .pth, for the host (PIL) and device (jpeg.hip + resample.hip) front ends."""
import os, sys, time, json, tempfile, pathlib, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oadp_amd import clip
from oadp_amd.config import Config
from oadp_amd.oake import globals as globals_, blocks, objects
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
only = sys.argv[3].split(',') if len(sys.argv) > 3 else ['globals', 'blocks', 'objects']
root = pathlib.Path(tempfile.mkdtemp(prefix='oake_sweep_'))
(root / 'images').mkdir()
rng = np.random.default_rng(0)
images = []
for i in range(n):
    w, h = (640, 480) if i % 3 else (480, 640)
    yy, xx = np.mgrid[0:h, 0:w]
    a = (rng.integers(0, 24, (h, w, 3)) + np.stack([(xx * 3 + yy + i) % 200, (xx + yy * 2) % 200, (xx * yy // 7) % 200], -1)).astype(np.uint8)
    name = f'{i:012d}.jpg'
    Image.fromarray(a).save(root / 'images' / name, quality=85, subsampling=2)
    images.append(dict(id=i, file_name=name, width=w, height=h))
(root / 'ann.json').write_text(json.dumps(dict(images=images, annotations=[], categories=[])))
# objects mode: 300 synthetic proposals per image (sorted ids), as BASELINE.json configs[3]
import pickle
props = []
for im in images:
    w, h = im['width'], im['height']
    x1 = rng.uniform(0, w * 0.7, 300); y1 = rng.uniform(0, h * 0.7, 300)
    bw = np.exp(rng.uniform(np.log(8), np.log(min(w, h)), 300)); bh = np.exp(rng.uniform(np.log(8), np.log(min(w, h)), 300))
    sc = np.sort(rng.uniform(0, 1, 300))[::-1]
    props.append(np.stack([x1, y1, np.minimum(x1 + bw, w), np.minimum(y1 + bh, h), sc], 1).astype(np.float32))
with open(root / 'props.pkl', 'wb') as f:
    pickle.dump(props, f)
sd = synthetic_state_dict()
for cls, tag, bs in ((globals_.Validator, 'globals', 256), (blocks.Validator, 'blocks', 1024)):
    if tag not in only:
        continue
    # (device decode through DataLoader workers is no longer a configuration: the validator switches the workers
    # off — it measured 117 vs 3 135 images/s in blocks mode, profiles/r02_sweep_1gpu.log)
    for mode, kw, nw in (('host PIL decode + PIL preprocess', {}, workers),
                         ('device decode + preprocess, no workers', dict(device_decode=True), 0)):
        out = root / f'{tag}_{len(kw)}_{nw}'
        model, pre = clip.load(sd, max_batch=512)
        model.encode_image(torch.zeros(2, 3, 224, 224, device='cuda'))  # handle + weights before the clock
        dl = Config(dataset=dict(root=str(root / 'images'), annFile=str(root / 'ann.json'), output_dir=str(out),
                                 transform=pre, **kw), num_workers=nw)
        v = cls(tag, model, dataloader=dl, device='cuda:0', batch_size=bs, log=dict(interval=10 ** 9),
                decode_threads=32)
        t0 = time.perf_counter(); c = v.run(); dt = time.perf_counter() - t0
        print(f'{tag:8s} {mode:40s}: {c.images} images, {c.crops} crops in {dt:.2f} s = {c.images/dt:.0f} images/s, {c.crops/dt:.0f} crops/s '
              f'({nw} workers)', flush=True)
n_obj = min(n, 96)  # 300 crops per image: 96 images = 28.8 k crops
for mode, kw, nw in (('host PIL decode + PIL preprocess', {}, workers), ('device decode + preprocess, no workers', dict(device_decode=True), 0)):
    if 'objects' not in only:
        break
    out = root / f'objects_{len(kw)}_{nw}'
    model, pre = clip.load(sd, max_batch=512)
    v_ = model.visual
    v_.positional_embedding = v_.interpolate_positional_embedding((14, 14)); v_.grid = 14
    v_.conv1.stride = (16, 16); v_.conv1.padding = (15, 15); v_.object_stream = True
    # handle, buffers and weight upload before the clock (as for the other two modes)
    v_(torch.zeros(2, 3, 224, 224, device='cuda', dtype=torch.float16), torch.zeros(2, 1, 14, 14, device='cuda', dtype=torch.float16))
    torch.cuda.synchronize()
    ann = json.loads((root / 'ann.json').read_text()); ann['images'] = ann['images'][:n_obj]
    (root / 'ann_obj.json').write_text(json.dumps(ann))
    with open(root / 'props_obj.pkl', 'wb') as f:
        pickle.dump(props[:n_obj], f)
    dl = Config(dataset=dict(type='COCODataset', root=str(root / 'images'), annFile=str(root / 'ann_obj.json'), output_dir=str(out),
                             transform=pre, grid=14, proposal_file=str(root / 'props_obj.pkl'), proposal_sorted=True, **kw), num_workers=nw)
    v = objects.Validator('objects', model, dataloader=dl, device='cuda:0', batch_size=1024, mini_batch_size=512,
                          log=dict(interval=10 ** 9), decode_threads=32)
    t0 = time.perf_counter(); c = v.run(); dt = time.perf_counter() - t0
    print(f'objects  {mode:40s}: {c.images} images, {c.crops} crops in {dt:.2f} s = {c.images/dt:.1f} images/s, {c.crops/dt:.0f} crops/s '
          f'({nw} workers)', flush=True)
shutil.rmtree(root)
