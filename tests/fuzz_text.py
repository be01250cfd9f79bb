"""Random text-tower configurations through oake_encode_text against the oracle (GPU box; hand-run): context
2..130 (the causal attention kernels for 1..130 keys, incl. last chunks of 49..63 keys), width 64..512, 1..3 layers,
1..400 sequences with max_batch below / above the batch, trimmed contexts.  usage: python tests/fuzz_text.py [n=60] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oadp_amd import clip
from oadp_amd.weights import synthetic_text_state_dict, synthetic_tokens
from oracle.text_ref import TextConfig, encode_text_ref
from oracle.vit_ref import l2_normalize

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device('cuda:0')
bad = 0
for it in range(n_cases):
    width = int(rng.choice([64, 128, 256, 512]))
    arch = dict(context=int(rng.integers(2, 131)), vocab=int(rng.integers(50, 3000)), width=width,
                layers=int(rng.integers(1, 4)), heads=width // 64, mlp_dim=64 * int(rng.integers(1, 4 * width // 64 + 1)),
                embed_dim=8 * int(rng.integers(1, 65)))
    n = int(rng.integers(1, max(2, min(400, 40000 // arch['context']))))
    length = arch['context'] if rng.random() < 0.5 else int(rng.integers(2, arch['context'] + 1))
    max_batch = int(rng.choice([max(1, n // 3), n, n + 3, 32]))
    dtype = torch.float16 if rng.random() < 0.7 else torch.bfloat16
    info = dict(arch, n=n, length=length, max_batch=max_batch, dtype=str(dtype).split('.')[-1])
    try:
        sd = synthetic_text_state_dict(seed=int(rng.integers(1, 1000)), **arch)
        model, _ = clip.load(sd, compute_dtype=dtype, max_batch=max_batch)
        tok = synthetic_tokens(n, length, arch['vocab'], seed=it)
        ref = l2_normalize(encode_text_ref(sd, TextConfig(**arch), tok))
        got = model.encode_text(tok.to(dev), normalize=True, out_dtype=torch.float32).cpu()
        tol = 1.5e-3 if dtype == torch.float16 else 2.5e-2
        err = ((got - ref).abs() - tol * ref.abs()).max().item()  # (atol + rtol, as the tests)
        cos = torch.nn.functional.cosine_similarity(got, ref, dim=1).min().item()
        if not torch.isfinite(got).all() or err > tol or cos < 0.999:
            bad += 1
            print('MISMATCH', info, 'max err', err, 'min cos', cos)
    except Exception as e:
        bad += 1
        print('RAISED', info, repr(e)[:300])
print(f'fuzz_text seed {seed}: {n_cases} random text-tower configurations, {bad} failures')
sys.exit(1 if bad else 0)
