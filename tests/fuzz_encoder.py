"""Random architectures and batch shapes through the whole encoder boundary against the fp32 oracle (GPU box;
hand-run, not collected by pytest; lives under tests/ because it drives oracle/): width 64..768, 1..3 layers,
patch 16 / 32, image 32..224, batches of 1..700 crops with max_batch below, at and above the batch (several
passes), f16 / bf16 operands, 16-bit and fp32 residual stream, CLS-only last block on and off, 16-bit and fp32
inputs (conv1 patch gather vs im2col), and the objects-mode dual stream after the reference's surgery with random
background masks.  usage: python tests/fuzz_encoder.py [n=60] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oadp_amd import clip
from oadp_amd.weights import synthetic_state_dict, normal
from oracle.vit_ref import ViTConfig, encode_image_ref, encode_objects_ref, l2_normalize

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device('cuda:0')
bad = 0
for it in range(n_cases):
    width = int(rng.choice([64, 128, 192, 256, 384, 512, 768]))
    layers = int(rng.integers(1, 4))
    patch = int(rng.choice([16, 32]))
    grid = int(rng.integers(1, 224 // patch + 1))
    image = grid * patch
    mlp = 64 * int(rng.integers(1, 4 * width // 64 + 1))
    embed = 8 * int(rng.integers(1, 65))
    arch = dict(image_size=image, patch_size=patch, width=width, layers=layers, heads=width // 64, mlp_dim=mlp, embed_dim=embed)
    objects_mode = rng.random() < 0.35 and grid >= 2
    tokens = grid * grid + 1 if not objects_mode else (2 * grid) ** 2 + 1
    n = int(rng.integers(1, max(2, min(700, 60000 // tokens))))
    max_batch = int(rng.choice([max(1, n // 3), n, n + 5, 256]))
    dtype = torch.float16 if rng.random() < 0.7 else torch.bfloat16
    resid32 = rng.random() < 0.25
    sd = synthetic_state_dict(seed=int(rng.integers(1, 1000)), **arch)
    info = dict(arch, n=n, max_batch=max_batch, dtype=str(dtype).split('.')[-1], resid32=bool(resid32), objects=bool(objects_mode))
    try:
        model, _ = clip.load(sd, compute_dtype=dtype, max_batch=max_batch, residual_dtype=torch.float32 if resid32 else None)
        x = normal(f'fuzz{it}', (n, 3, image, image), seed=seed)
        in16 = rng.random() < 0.5
        xin = (x.to(dtype) if in16 else x).to(dev)
        xref = x.to(dtype).float() if in16 else x
        tol = 1.5e-3 if dtype == torch.float16 else 2.5e-2
        if objects_mode:
            v = model.visual
            v.positional_embedding = v.interpolate_positional_embedding((v.grid * 2,) * 2)
            v.grid *= 2
            v.conv1.stride = tuple(s // 2 for s in v.conv1.stride)
            v.conv1.padding = ((v.patch_size - 1) // 2,) * 2
            v.object_stream = True
            g2 = v.grid
            masks = (torch.from_numpy(rng.random((n, 1, g2, g2))) < 0.6).float()
            sd2 = dict(sd); sd2['visual.positional_embedding'] = v.positional_embedding.detach().cpu().float()
            cfg = ViTConfig(stride=patch // 2, padding=(patch - 1) // 2, **arch)
            ref = l2_normalize(encode_objects_ref(sd2, cfg, xref, masks))
            got = v(xin, masks.to(dev).to(dtype if rng.random() < 0.5 else torch.float32), normalize=True, out_dtype=torch.float32)
        else:
            if rng.random() < 0.3:
                model.visual.set_option('cls_last', 0)
            ref = l2_normalize(encode_image_ref(sd, ViTConfig(stride=patch, **arch), xref))
            got = model.encode_image(xin, normalize=True, out_dtype=torch.float32)
        got = got.cpu()
        cos = torch.nn.functional.cosine_similarity(got, ref, dim=1).min().item()
        err = ((got - ref).abs() - tol * ref.abs()).max().item()  # (atol + rtol, as the tests)
        if not torch.isfinite(got).all() or err > tol or cos < 0.999:
            bad += 1
            print('MISMATCH', info, 'max err', err, 'min cos', cos)
    except Exception as e:
        bad += 1
        print('RAISED', info, repr(e)[:300])
print(f'fuzz_encoder seed {seed}: {n_cases} random encoder configurations, {bad} failures')
sys.exit(1 if bad else 0)
