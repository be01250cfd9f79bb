"""Randomised device-vs-Pillow check of the crop / resize / normalise path (GPU box): random source sizes
(1..1600, with a share of extreme aspect ratios), random float boxes (inside, across borders, far outside, a
few pixels small, larger than the image), squash and Resize+CenterCrop, plus random whole-image resizes
(the pyramid step).  Everything must equal PIL + the torchvision-style transform bit for bit.
usage: resample_fuzz.py [n_images=300] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import PIL.Image
from oadp_amd import clip
from oadp_amd.clip.preprocess import Preprocess
from oadp_amd.weights import synthetic_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
OUT = int(os.environ.get('OUT_SIZE', 224))  # the model's input resolution = the crop size (a multiple of 32)
model, _ = clip.load(synthetic_state_dict(image_size=OUT, patch_size=32, width=128, layers=1, heads=2, mlp_dim=256,
                                          embed_dim=64), max_batch=2)
vis = model.visual
# objects-mode twin (conv1 stride // 2, padding (patch - 1) // 2): its f16 crops go straight into the zero-padded batch
# conv1 gathers from (resample_v4p_kernel, OAKE_LAYOUT_PADDED) and come back as a strided view of that pool — compared
# with the dense f16 crops of the plain model, bit for bit, and the pool's border must stay zero
omodel, _ = clip.load(synthetic_state_dict(image_size=OUT, patch_size=32, width=128, layers=1, heads=2, mlp_dim=256,
                                           embed_dim=64), max_batch=2)
ovis = omodel.visual
ovis.positional_embedding = ovis.interpolate_positional_embedding((ovis.grid * 2,) * 2)
ovis.grid *= 2
ovis.conv1.stride = (16, 16)
ovis.conv1.padding = (15, 15)
ovis.object_stream = True
dev = torch.device('cuda:0')
bad = crops = resizes = padded = 0
for it in range(n):
    r = rng.random()
    if r < 0.15:
        w, h = int(rng.integers(1, 12)), int(rng.integers(1, 1600))
    elif r < 0.3:
        w, h = int(rng.integers(1, 1600)), int(rng.integers(1, 12))
    else:
        w, h = int(rng.integers(1, 1600)), int(rng.integers(1, 1200))
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if rng.random() < 0.5:
        yy, xx = np.mgrid[0:h, 0:w]
        a[..., 0] = ((xx * 255) // max(w - 1, 1)).astype(np.uint8)
        a[: h // 3, : w // 3] = 255
    pil = PIL.Image.fromarray(a)
    d = torch.from_numpy(a).to(dev)
    boxes = [(0.0, 0.0, float(w), float(h))]
    for _ in range(int(rng.integers(1, 9))):
        k = rng.random()
        if k < 0.5:
            x1, y1 = rng.uniform(0, w), rng.uniform(0, h)
            x2, y2 = x1 + rng.uniform(0.6, w), y1 + rng.uniform(0.6, h)
        elif k < 0.8:
            x1, y1 = rng.uniform(-w, w), rng.uniform(-h, h)
            x2, y2 = x1 + rng.uniform(1, 2 * w + 2), y1 + rng.uniform(1, 2 * h + 2)
        else:
            x1, y1 = rng.uniform(0, w), rng.uniform(0, h)
            x2, y2 = x1 + rng.uniform(0.6, 6), y1 + rng.uniform(0.6, 6)
        if round(x2) - round(x1) < 1 or round(y2) - round(y1) < 1:
            continue
        boxes.append((float(np.float32(x1)), float(np.float32(y1)), float(np.float32(x2)), float(np.float32(y2))))
    for squash in (False, True):
        host = Preprocess(OUT, squash=squash)
        try:
            out = vis.crop_resize_normalize_batch([d], [boxes], squash=squash, out_dtype=torch.float32)
        except Exception as e:  # (e.g. a Resize that would exceed the supported size: must be loud, not wrong)
            print('RAISED', (w, h), squash, str(e)[:120])
            continue
        o16 = vis.crop_resize_normalize_batch([d], [boxes], squash=squash, out_dtype=torch.float16)
        p16 = ovis.crop_resize_normalize_batch([d], [boxes], squash=squash, out_dtype=torch.float16)
        if p16.is_contiguous():
            print('PADDED POOL NOT USED')
            bad += 1
        padded += p16.shape[0]
        if not torch.equal(p16, o16):
            bad += 1
            print('MISMATCH padded crops', (w, h), 'squash', squash, int((p16 != o16).sum()))
        for b, o in zip(boxes, out):
            crops += 1
            try:
                ref = host(pil.crop(b))
            except Exception as e:
                print('PIL raised', (w, h), b, e)
                continue
            if not torch.equal(o.cpu(), ref):
                bad += 1
                dd = (o.cpu() - ref).abs()
                print('MISMATCH crop', (w, h), b, 'squash', squash, 'max', float(dd.max()), 'n', int((dd > 0).sum()))
    ow, oh = max(1, int(w / rng.uniform(1.0, 3.0))), max(1, int(h / rng.uniform(1.0, 3.0)))
    if rng.random() < 0.2:
        ow, oh = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    got = vis.resize_u8(d, (ow, oh)).cpu().numpy()
    ref = np.asarray(pil.resize((ow, oh), PIL.Image.BICUBIC))
    resizes += 1
    if not np.array_equal(got, ref):
        bad += 1
        print('MISMATCH resize', (w, h), '->', (ow, oh), 'max', int(np.abs(got.astype(int) - ref).max()))
for pool, pad, hp, ws in ovis._pad_pools.values():
    border = pool.clone()
    border[:, :, pad:pad + OUT, pad:pad + OUT] = 0
    if border.any():
        bad += 1
        print('PADDED POOL: border not zero', int((border != 0).sum()))
print(f'resample_fuzz seed {seed} (out {OUT}): {n} images, {crops} crops and {resizes} resizes compared with PIL, '
      f'{padded} crops through the zero-padded batch compared with the dense ones, {bad} mismatches')
sys.exit(1 if bad else 0)
