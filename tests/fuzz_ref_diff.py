"""Differential run of the product's host-side crop math against the REFERENCE's own functions (build container
only: needs /root/reference; the reference modules are loaded from where they lie under the import stubs of
tools/gen_golden.py, nothing of them is copied).  Far more cases than the committed goldens:
  _partition for every length 0..6000; pyramid tiles + bboxes for random image sizes up to 4000 px;
  _mask for thousands of random (foreground, object) pairs incl. fractional and degenerate ones;
  _expand + the crop boxes PIL derives from it for random proposal sets on random image sizes.
Integer / index results must be identical; the float boxes of _expand to 1e-3 px (torch's vectorised sqrt differs
by an ulp between code paths) with identical PIL crop boxes.  (Lives under tests/ because it also drives oracle/; not collected by pytest: run it by hand.)
usage: python tests/fuzz_ref_diff.py [seed=0]"""
import pathlib, sys, types
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1] / 'tools'))
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, PIL.Image, torch
import gen_golden as gg
from oadp_amd.clip.preprocess import Preprocess
from oadp_amd.oake import blocks as pblocks, objects as pobjects
from oracle import crops_ref

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
gg.install_stubs()
gg.load_ref('base')
rb, ro = gg.load_ref('blocks'), gg.load_ref('objects')
bad = 0

# ---- blocks ----
rds = rb.Dataset.__new__(rb.Dataset); rds._r, rds._s, rds._rescale = 224, 112, 1.5
pds = pblocks.Dataset.__new__(pblocks.Dataset); pds._r, pds._s, pds._rescale = 224, 112, 1.5
for n in range(0, 6001):
    a, b = rds._partition(n), pds._partition(n)
    if a != b or a != crops_ref.partition(n):
        bad += 1; print('PARTITION', n, a, b)
n_sizes = 0
for _ in range(400):
    w, h = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
    img = PIL.Image.new('RGB', (w, h))
    ta = [[im.size[0], im.size[1], sc, x, y] for im, sc, x, y in rds._partitions(img)]
    tb = [[im.size[0], im.size[1], sc, x, y] for im, sc, x, y in pds._partitions(img)]
    ba = [tuple(rds._bbox(sc, x, y)) for _, _, sc, x, y in ta]
    bb = [tuple(pds._bbox(sc, x, y)) for _, _, sc, x, y in tb]
    if ta != tb or ba != bb:
        bad += 1; print('TILES', (w, h), len(ta), len(tb))
    n_sizes += 1

# ---- objects: masks ----
rod = ro.COCODataset.__new__(ro.COCODataset); rod._grid = 14; rod._expand_mode = ro.ExpandMode.ADAPTIVE
pod = pobjects.COCODataset.__new__(pobjects.COCODataset); pod._grid = 14; pod._expand_mode = pobjects.ExpandMode.ADAPTIVE
n_masks = 0
fgs, obs = [], []
for _ in range(3000):
    ow, oh = int(rng.integers(1, 700)), int(rng.integers(1, 700))
    if rng.random() < 0.5:
        oh = ow
    x1, y1 = rng.uniform(-3, ow), rng.uniform(-3, oh)
    x2, y2 = x1 + rng.uniform(0.0, ow + 3), y1 + rng.uniform(0.0, oh + 3)
    ox, oy = float(rng.integers(0, 50)), float(rng.integers(0, 50))
    if rng.random() < 0.3:
        ox += float(rng.random()); oy += float(rng.random())
    fg = (float(np.float32(x1)), float(np.float32(y1)), float(np.float32(x2)), float(np.float32(y2)))
    ob = (ox, oy, ox + ow, oy + oh)
    ma = rod._mask(fg, ob).reshape(14, 14)
    mb = pod._mask(fg, ob).reshape(14, 14)
    if not torch.equal(ma, mb) or not np.array_equal(crops_ref.object_mask(fg, ob), ma.numpy().astype(np.uint8)):
        bad += 1; print('MASK', fg, ob)
    fgs.append(fg); obs.append(ob); n_masks += 1
# the vectorised form the product actually runs, on the same cases
# (boxes as the reference's loop sees them: float32 tensors iterated into tuples of Python floats — todd.BBox —
# so differences are taken in double precision on float32-rounded values)
fg32, ob32 = torch.tensor(fgs, dtype=torch.float32), torch.tensor(obs, dtype=torch.float32)
mv = pod._masks(fg32, ob32)
for i in range(len(fgs)):
    if not torch.equal(mv[i, 0], rod._mask(tuple(fg32[i].tolist()), tuple(ob32[i].tolist())).reshape(14, 14)):
        bad += 1; print('MASKS(vectorised)', fgs[i], obs[i])

# ---- objects: expand ----
n_exp = 0
for _ in range(300):
    w, h = int(rng.integers(8, 2000)), int(rng.integers(8, 2000))
    k = int(rng.integers(1, 60))
    cx, cy = rng.uniform(0, w, k), rng.uniform(0, h, k)
    bw = np.exp(rng.uniform(np.log(1), np.log(max(w, h) * 1.5), k)); bh = np.exp(rng.uniform(np.log(1), np.log(max(w, h) * 1.5), k))
    prop = np.stack([np.clip(cx - bw / 2, 0, w), np.clip(cy - bh / 2, 0, h), np.clip(cx + bw / 2, 0, w), np.clip(cy + bh / 2, 0, h)], 1).astype(np.float32)
    p_ = gg.BBoxesXYXY(torch.tensor(prop))
    keep = p_.indices(min_wh=(4, 4))
    if keep.tolist() != pobjects.indices_min_wh(torch.tensor(prop), (4, 4)).tolist():
        bad += 1; print('KEEP', (w, h))
    if not keep.any():
        continue
    ea = rod._expand(p_[keep], torch.tensor([w, h])).to_tensor().numpy()
    eb = pod._expand(torch.tensor(prop)[keep], torch.tensor([w, h])).numpy()
    ec = crops_ref.expand_adaptive(prop[keep.numpy()], (w, h))
    if not (np.allclose(ea, eb, rtol=0, atol=1e-3) and np.allclose(ea, ec, rtol=0, atol=1e-3)):
        bad += 1; print('EXPAND', (w, h), np.abs(ea - eb).max(), np.abs(ea - ec).max())
    for a, b in zip(ea, eb):
        if crops_ref.pil_crop_box(a) != crops_ref.pil_crop_box(b):
            # (a float box within an ulp of a .5 boundary may round differently: count, do not fail silently)
            bad += 1; print('CROPBOX', (w, h), a, b)
    n_exp += int(keep.sum())
print(f'fuzz_ref_diff seed {seed}: 6001 partition lengths, {n_sizes} pyramid sizes, {n_masks} masks (+ vectorised), '
      f'{n_exp} expanded boxes: {bad} mismatches')
sys.exit(1 if bad else 0)
