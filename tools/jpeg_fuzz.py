"""Randomised device-vs-Pillow JPEG decode check (GPU box): N synthetic images of random size, content,
quality, chroma subsampling, restart interval, Huffman-table optimisation and progressive flag, encoded by
Pillow, decoded by oake_decode_jpeg_batch and by PIL.Image.open().convert('RGB'); every pixel must agree
(files the device decoder declines — CMYK, 12-bit, arithmetic — must come back as None, never as wrong pixels).
usage: jpeg_fuzz.py [n=2000] [seed=0]"""
import io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oadp_amd import clip
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
model, _ = clip.load(synthetic_state_dict(image_size=64, patch_size=32, width=128, layers=1, heads=2, mlp_dim=256,
                                          embed_dim=64), max_batch=2)
bad = declined = checked = 0
batch, metas = [], []
def flush():
    global bad, declined, checked
    outs = model.visual.decode_jpeg_batch(batch, threads=16)
    for data, meta, o in zip(batch, metas, outs):
        ref = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
        if o is None:
            declined += 1
            if meta['mode'] == 'RGB' or meta['mode'] == 'L':
                print('DECLINED', meta)
            continue
        checked += 1
        if not np.array_equal(o.cpu().numpy(), ref):
            bad += 1
            d = np.abs(o.cpu().numpy().astype(int) - ref.astype(int))
            print('MISMATCH', meta, 'max diff', d.max(), 'pixels', int((d > 0).any(-1).sum()))
    batch.clear(); metas.clear()
for i in range(n):
    h, w = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    if rng.random() < 0.1:
        h, w = int(rng.integers(1, 20)), int(rng.integers(1, 20))
    kind = rng.integers(0, 4)
    if kind == 0:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    elif kind == 1:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 5) % 256], -1).astype(np.uint8)
    elif kind == 2:
        a = np.full((h, w, 3), rng.integers(0, 256, 3), dtype=np.uint8)
        a[h // 3:, w // 2:] = rng.integers(0, 256, 3)
    else:
        a = (rng.integers(0, 32, (h, w, 3)) + np.linspace(0, 220, w)[None, :, None]).astype(np.uint8)
    mode = 'L' if rng.random() < 0.12 else 'RGB'
    img = Image.fromarray(a).convert(mode)
    kw = dict(quality=int(rng.integers(1, 101)), optimize=bool(rng.random() < 0.4), progressive=bool(rng.random() < 0.2))
    if mode == 'RGB':
        kw['subsampling'] = int(rng.integers(0, 3))
    if rng.random() < 0.25:
        kw['restart_marker_blocks'] = int(rng.integers(1, 40))
    elif rng.random() < 0.1:
        kw['restart_marker_rows'] = int(rng.integers(1, 5))
    buf = io.BytesIO()
    try:
        img.save(buf, 'JPEG', **kw)
    except OSError:  # (Pillow's encoder refuses a few parameter combinations)
        continue
    batch.append(buf.getvalue()); metas.append(dict(i=i, h=h, w=w, mode=mode, **kw))
    if len(batch) == 64:
        flush()
if batch:
    flush()
print(f'jpeg_fuzz seed {seed}: {n} files, {checked} decoded on the device and compared, {declined} declined, {bad} mismatches')
sys.exit(1 if bad else 0)
