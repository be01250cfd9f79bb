"""The OAKE sweep (files -> .pth, device front end) with R processes sharing ONE GPU: each rank is a full
validator on its DistributedSampler shard of the images, so the per-image host work (file reads, Huffman
decode, index math, .pth writing — one interpreter each) scales with R while the GPU interleaves their
kernels.  Launched under torch.distributed.run; gloo carries the one counters gather.  GPU box only.
usage: python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 --master-port P \
           tools/sweep_ranks.py [n_images=4096] [globals|blocks|objects] [dataset dir]"""
import json, os, pathlib, pickle, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as td
from PIL import Image
from oadp_amd import clip
from oadp_amd.config import Config
from oadp_amd.oake import globals as globals_, blocks, objects
from oadp_amd.oake.base import gather_counters
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else 'blocks'
rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
if world > 1:
    td.init_process_group(backend='gloo')
root = pathlib.Path(sys.argv[3] if len(sys.argv) > 3 else os.path.join(tempfile.gettempdir(), f'oake_ranks_{n}'))
def _make(args):
    path, i = args
    w, h = (640, 480) if i % 3 else (480, 640)
    rng = np.random.default_rng(i)
    yy, xx = np.mgrid[0:h, 0:w]
    a = (rng.integers(0, 24, (h, w, 3)) + np.stack([(xx * 3 + yy + i) % 200, (xx + yy * 2) % 200, (xx * yy // 7) % 200], -1)).astype(np.uint8)
    Image.fromarray(a).save(path, quality=85, subsampling=2)


if rank == 0 and not (root / 'ann.json').exists():
    from multiprocessing import Pool
    (root / 'images').mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(0)
    images, props = [], []
    with Pool(min(64, os.cpu_count() or 8)) as pool:  # (before any HIP call of this process: fork is safe)
        pool.map(_make, [(str(root / 'images' / f'{i:012d}.jpg'), i) for i in range(n)], chunksize=64)
    for i in range(n):
        w, h = (640, 480) if i % 3 else (480, 640)
        images.append(dict(id=i, file_name=f'{i:012d}.jpg', width=w, height=h))
        x1 = rng.uniform(0, w * 0.7, 300); y1 = rng.uniform(0, h * 0.7, 300)
        bw = np.exp(rng.uniform(np.log(8), np.log(min(w, h)), 300)); bh = np.exp(rng.uniform(np.log(8), np.log(min(w, h)), 300))
        sc = np.sort(rng.uniform(0, 1, 300))[::-1]
        props.append(np.stack([x1, y1, np.minimum(x1 + bw, w), np.minimum(y1 + bh, h), sc], 1).astype(np.float32))
    with open(root / 'props.pkl', 'wb') as f:
        pickle.dump(props, f)
    (root / 'ann.json').write_text(json.dumps(dict(images=images, annotations=[], categories=[])))
if world > 1:
    td.barrier()
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
out = root / f'out_{which}_{world}'
if rank == 0:
    shutil.rmtree(out, ignore_errors=True)
if world > 1:
    td.barrier()
model, pre = clip.load(synthetic_state_dict(), max_batch=512)
ds = dict(root=str(root / 'images'), annFile=str(root / 'ann.json'), output_dir=str(out), transform=pre, device_decode=True)
if which == 'objects':
    v_ = model.visual
    v_.positional_embedding = v_.interpolate_positional_embedding((14, 14)); v_.grid = 14
    v_.conv1.stride = (16, 16); v_.conv1.padding = (15, 15); v_.object_stream = True
    v_(torch.zeros(2, 3, 224, 224, device='cuda', dtype=torch.float16), torch.zeros(2, 1, 14, 14, device='cuda', dtype=torch.float16))
    ds.update(type='COCODataset', grid=14, proposal_file=str(root / 'props.pkl'), proposal_sorted=True)
    v = objects.Validator('objects', model, dataloader=Config(dataset=ds, num_workers=0), device='cuda', batch_size=1024,
                          mini_batch_size=512, log=dict(interval=10 ** 9), decode_threads=int(os.environ.get('DECODE_THREADS', max(4, 32 // world))),
                          host_threads=int(os.environ.get('HOST_THREADS', 8)))
else:
    model.encode_image(torch.zeros(2, 3, 224, 224, device='cuda'))
    cls, bs = (globals_.Validator, 256) if which == 'globals' else (blocks.Validator, 1024)
    v = cls(which, model, dataloader=Config(dataset=ds, num_workers=0), device='cuda', batch_size=bs,
            log=dict(interval=10 ** 9), decode_threads=int(os.environ.get('DECODE_THREADS', max(4, 32 // world))),
            host_threads=int(os.environ.get('HOST_THREADS', 8)))
torch.cuda.synchronize()
if world > 1:
    td.barrier()
# host load while the sweep runs (rank 0 samples the whole box: the question is how many ranks the HOST carries)
cpu_samples, stop = [], [False]
if rank == 0:
    import threading, psutil
    def _sample():
        psutil.cpu_percent(None)
        while not stop[0]:
            cpu_samples.append(psutil.cpu_percent(0.25))
    threading.Thread(target=_sample, daemon=True).start()
t0 = time.perf_counter()
c = v.run()
torch.cuda.synchronize()
if world > 1:
    td.barrier()
dt = time.perf_counter() - t0
stop[0] = True
per_rank = gather_counters(c, torch.device('cpu'))
if rank == 0:
    images = sum(r[0] for r in per_rank); crops = sum(r[1] for r in per_rank)
    print(f'{which:8s} {world} rank(s) on one GPU: {int(images)} images, {int(crops)} crops in {dt:.2f} s = '
          f'{images / dt:.0f} images/s, {crops / dt:.0f} crops/s; host CPU {sum(cpu_samples) / max(len(cpu_samples), 1):.1f} % of '
          f'{os.cpu_count()} logical cores (mean of {len(cpu_samples)} samples)', flush=True)
if world > 1:
    td.destroy_process_group()
