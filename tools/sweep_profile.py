"""cProfile of the main thread of the globals sweep with the device front end (GPU box): where the
per-image host time goes.  usage: sweep_profile.py [n_images] [globals|blocks]"""
import cProfile, io, json, os, pathlib, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oadp_amd import clip
from oadp_amd.config import Config
from oadp_amd.oake import globals as globals_, blocks
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
which = sys.argv[2] if len(sys.argv) > 2 else 'globals'
root = pathlib.Path(tempfile.mkdtemp(prefix='oake_prof_'))
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
model, pre = clip.load(synthetic_state_dict(), max_batch=512)
model.encode_image(torch.zeros(2, 3, 224, 224, device='cuda'))
cls, bs = (globals_.Validator, 256) if which == 'globals' else (blocks.Validator, 1024)
dl = Config(dataset=dict(root=str(root / 'images'), annFile=str(root / 'ann.json'), output_dir=str(root / 'out'),
                         transform=pre, device_decode=True), num_workers=0)
v = cls(which, model, dataloader=dl, device='cuda:0', batch_size=bs, log=dict(interval=10 ** 9),
        decode_threads=int(os.environ.get('DECODE_THREADS', 32)))
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable(); c = v.run(); pr.disable(); dt = time.perf_counter() - t0
print(f'{which}: {c.images} images, {c.crops} crops in {dt:.2f} s = {c.images / dt:.0f} images/s')
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:6000])
shutil.rmtree(root)
