"""Randomised check of blocks mode's device path (GPU box): random image sizes (1..1500 px, some below one
block, some exactly on the partition edges), `oake_blocks_batch` over flushes of several images against the host
dataset's `_preprocess` (PIL pyramid + crops, the reference's formulation): block counts, bboxes order and every
pixel of every crop, bit for bit.  usage: blocks_fuzz.py [n_images=80] [seed=0]"""
import os, pathlib, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import PIL.Image
from oadp_amd import clip
from oadp_amd.oake import blocks
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
R, S, RESCALE = (lambda t: (int(t[0]), int(t[1]), float(t[2])))(os.environ.get('BLOCK', '224,112,1.5').split(','))
model, pre = clip.load(synthetic_state_dict(image_size=R, patch_size=32, width=128, layers=1, heads=2, mlp_dim=256,
                                            embed_dim=64), max_batch=2)
ds = blocks.Dataset.__new__(blocks.Dataset)
ds._r, ds._s, ds._rescale = R, S, RESCALE
ds.transform = pre
dev = torch.device('cuda:0')
EDGE = [R - 1, R, R + 1, R + S - 1, R + S, R + S + 1, 2 * R, 2 * R + 1, int(R * RESCALE), int(R * RESCALE) + 1]
bad = crops = 0
i = 0
while i < n:
    k = int(rng.integers(1, 6))
    arrs = []
    for _ in range(k):
        w = int(rng.choice(EDGE)) if rng.random() < 0.25 else int(rng.integers(1, 1500))
        h = int(rng.choice(EDGE)) if rng.random() < 0.25 else int(rng.integers(1, 1100))
        arrs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    out, counts = model.visual.blocks_batch([torch.from_numpy(a).to(dev) for a in arrs], block_size=R,
                                            max_stride=S, rescale=RESCALE, out_dtype=torch.float32)
    out = out.cpu()
    i0 = 0
    for a, c in zip(arrs, counts):
        host = ds._preprocess(0, pathlib.Path('x'), PIL.Image.fromarray(a))
        kk = host.blocks.shape[0]
        if kk != c:
            bad += 1
            print('COUNT', a.shape, c, kk)
        elif not torch.equal(out[i0:i0 + c], host.blocks):
            bad += 1
            d = (out[i0:i0 + c] - host.blocks).abs().flatten(1).max(1).values
            print('MISMATCH', a.shape, 'blocks', torch.nonzero(d > 0).flatten().tolist()[:8], 'of', c)
        crops += c
        i0 += c
    i += k
print(f'blocks_fuzz seed {seed} (block {R}, stride {S}, rescale {RESCALE}): {i} images, {crops} crops compared with the host (PIL) dataset, {bad} mismatches')
sys.exit(1 if bad else 0)
