import numpy as np, torch, PIL.Image, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oadp_amd import clip
from oadp_amd.weights import synthetic_state_dict
model, _ = clip.load(synthetic_state_dict(width=128, layers=1, heads=2, mlp_dim=256, embed_dim=64), max_batch=2)
v = model.visual
rng = np.random.default_rng(0)
for (w, h, ow, oh) in [(640, 480, 426, 320), (64, 48, 40, 32), (33, 20, 17, 9)]:
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = v.resize_u8(torch.from_numpy(a).cuda(), (ow, oh)).cpu().numpy()
    ref = np.asarray(PIL.Image.fromarray(a).resize((ow, oh), PIL.Image.BICUBIC))
    bad = np.argwhere(got != ref)
    print((w, h, ow, oh), 'mismatches', len(bad), 'rows', sorted(set(bad[:, 0]))[:10], 'cols', sorted(set(bad[:, 1]))[:20], 'chan', sorted(set(bad[:, 2])))
    if len(bad):
        r, c, ch = bad[0]
        print(' first', bad[0], 'got', got[r, max(0,c-2):c+6].tolist(), 'ref', ref[r, max(0,c-2):c+6].tolist())
