"""End-to-end OAKE sweep throughput on a synthetic COCO-like JPEG set (GPU box): files -> decode ->
preprocess -> encoder -> .pth, for the host (PIL) and device (jpeg.hip + resample.hip) front ends."""
import os, sys, time, json, tempfile, pathlib, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oadp_amd import clip
from oadp_amd.config import Config
from oadp_amd.oake import globals as globals_, blocks
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
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
sd = synthetic_state_dict()
for cls, tag, bs in ((globals_.Validator, 'globals', 256), (blocks.Validator, 'blocks', 1024)):
    for mode, kw, nw in (('host PIL decode + PIL preprocess', {}, workers),
                         ('device decode + preprocess, DataLoader workers', dict(device_decode=True), workers),
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
shutil.rmtree(root)
