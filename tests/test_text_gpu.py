"""Text tower (oadp/prompts/vild.py): oake_encode_text through the C ABI vs the oracle."""
import pytest
import torch

from oadp_amd import clip
from oadp_amd.weights import synthetic_text_state_dict, synthetic_tokens
from oracle.text_ref import TextConfig, encode_text_ref
from oracle.vit_ref import l2_normalize

pytestmark = pytest.mark.gpu

TINY = dict(context=24, vocab=500, width=128, layers=2, heads=2, mlp_dim=256, embed_dim=64)


def _check(out, ref, tol):
    cos = torch.nn.functional.cosine_similarity(out.float().cpu(), ref, dim=1)
    print(f'max|err|={(out.float().cpu() - ref).abs().max().item():.3e} min cos={cos.min().item():.6f}')
    assert cos.min().item() >= 0.999
    torch.testing.assert_close(out.float().cpu(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('n,length', [(1, 24), (7, 24), (5, 9), (70, 24), (300, 9)])
def test_encode_text_tiny(cuda, dtype, tol, n, length):
    """Small and large batches (70 x 24 = 1680 rows run the persistent GEMMs), trimmed contexts;
    (300, 9): more sequences than max_batch AND a trimmed context, so one pass packs 85 > max_batch = 32
    sequences — the head buffers (gather_eot -> ln_final -> projection) must hold them all."""
    sd = synthetic_text_state_dict(**TINY)
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=32)  # 70 > max_batch: multi-pass
    tok = synthetic_tokens(n, length, TINY['vocab'], seed=n)
    ref = l2_normalize(encode_text_ref(sd, TextConfig(**TINY), tok))
    out = model.encode_text(tok.to(cuda), normalize=True, out_dtype=torch.float32)
    _check(out, ref, tol)


def test_narrow_tower_with_a_wide_embedding(cuda):
    """width 64, embed 464: text_projection (64 x 464) is the largest tensor that goes through the fp32 staging
    buffer of the weight upload — it used to be sized for conv1 / MLP / in_proj / positional embedding only
    (found by tests/fuzz_text.py)."""
    arch = dict(context=20, vocab=807, width=64, layers=1, heads=1, mlp_dim=64, embed_dim=464)
    sd = synthetic_text_state_dict(**arch)
    model, _ = clip.load(sd, max_batch=16)
    tok = synthetic_tokens(21, 5, arch['vocab'], seed=3)
    ref = l2_normalize(encode_text_ref(sd, TextConfig(**arch), tok))
    _check(model.encode_text(tok.to(cuda), normalize=True, out_dtype=torch.float32), ref, 1e-3)


def test_encode_text_clip_b32_text_tower(cuda):
    """The real text architecture (77 x 512, 8 heads, 12 layers, vocab 49408) at 30 prompts = 2310 rows."""
    sd = synthetic_text_state_dict()
    model, _ = clip.load(sd, max_batch=64)
    tok = synthetic_tokens(30, 77, seed=4)
    ref = l2_normalize(encode_text_ref(sd, TextConfig(), tok))
    out = model.encode_text(tok.to(cuda), normalize=True, out_dtype=torch.float32)
    _check(out, ref, 1e-3)
    # causal model: trimming the padded tail of the context must not change anything
    longest = int(tok.argmax(dim=-1).max()) + 1
    out2 = model.encode_text(tok[:, :longest].contiguous().to(cuda), normalize=True, out_dtype=torch.float32)
    _check(out2, ref, 1e-3)
    with pytest.raises(RuntimeError):
        model.encode_text(tok)  # CPU tensor: no CPU path


def test_vild_prompt_ensemble(cuda, tmp_path):
    """oadp_amd.prompts.vild.main == the reference flow (normalise per template, average, save)
    computed with the oracle."""
    from oadp_amd.prompts import vild
    arch = dict(TINY, vocab=1200, context=16)
    sd = synthetic_text_state_dict(**arch)
    model, _ = clip.load(sd, max_batch=64)
    names = ['zebra', 'traffic light', 'hot dog', 'cat']
    enc = lambda s: [1 + (sum(map(ord, w)) % 1000) for w in s.split()]
    prompts = vild.templates()[:5]
    kw = dict(sot=1198, eot=1199, context=16)
    state = vild.main(names, enc, model=model, output=str(tmp_path / 'p' / 'vild.pth'), prompts=prompts, **kw)
    assert state['names'] == sorted(names)
    ref = sum(l2_normalize(encode_text_ref(sd, TextConfig(**arch),
                                           vild.adaptively_tokenize(map(p.format, sorted(names)), enc, **kw)))
              for p in prompts) / len(prompts)
    saved = torch.load(tmp_path / 'p' / 'vild.pth', 'cpu')
    assert saved['names'] == sorted(names)
    torch.testing.assert_close(saved['embeddings'], ref, rtol=1e-3, atol=1e-3)
