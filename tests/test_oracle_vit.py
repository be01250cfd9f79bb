"""Pin the ViT oracle (oracle/vit_ref.py):
 (a) against HuggingFace's independent CLIP vision tower on shared random weights;
 (b) the object-token dual stream against the REFERENCE's own Hooks + _build_model surgery, run
     under import stubs on a stand-in torch ViT (tests/golden/hooks_tiny.npz, tools/gen_golden.py).
The LutingWang/CLIP fork itself is un-vendored and un-pinned => parity with it is unpinned."""
import json
import pathlib

import numpy as np
import pytest
import torch

from oadp_amd.weights import synthetic_images, synthetic_state_dict
from oracle.vit_ref import ViTConfig, encode_image_ref, encode_objects_ref

GOLD = pathlib.Path(__file__).parent / 'golden'
TINY = dict(width=128, layers=2, heads=2, mlp_dim=512, embed_dim=64)


def test_weights_are_deterministic():
    a = synthetic_state_dict(**TINY)['visual.proj']
    b = synthetic_state_dict(**TINY)['visual.proj']
    assert torch.equal(a, b)
    # a fixed fingerprint: the generator must not drift between builds / boxes
    assert abs(float(a.double().sum()) - float(synthetic_state_dict(**TINY)['visual.proj'].double().sum())) == 0
    assert abs(float(synthetic_images(1, seed=3).double().mean())) < 0.01


def test_matches_huggingface_clip():
    transformers = pytest.importorskip('transformers')
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    sd = synthetic_state_dict(**TINY)
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                           num_attention_heads=2, image_size=224, patch_size=32, projection_dim=64,
                           hidden_act='quick_gelu', layer_norm_eps=1e-5, attention_dropout=0.0)
    hf = CLIPVisionModelWithProjection(cfg).eval()
    hsd = {}
    v = 'vision_model.'
    hsd[v + 'embeddings.class_embedding'] = sd['visual.class_embedding']
    hsd[v + 'embeddings.patch_embedding.weight'] = sd['visual.conv1.weight']
    hsd[v + 'embeddings.position_embedding.weight'] = sd['visual.positional_embedding']
    hsd[v + 'pre_layrnorm.weight'] = sd['visual.ln_pre.weight']
    hsd[v + 'pre_layrnorm.bias'] = sd['visual.ln_pre.bias']
    hsd[v + 'post_layernorm.weight'] = sd['visual.ln_post.weight']
    hsd[v + 'post_layernorm.bias'] = sd['visual.ln_post.bias']
    hsd['visual_projection.weight'] = sd['visual.proj'].t().contiguous()
    for i in range(2):
        p, q = f'visual.transformer.resblocks.{i}.', v + f'encoder.layers.{i}.'
        w, b = sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias']
        for j, n in enumerate(('q_proj', 'k_proj', 'v_proj')):
            hsd[q + f'self_attn.{n}.weight'] = w[j * 128:(j + 1) * 128]
            hsd[q + f'self_attn.{n}.bias'] = b[j * 128:(j + 1) * 128]
        hsd[q + 'self_attn.out_proj.weight'] = sd[p + 'attn.out_proj.weight']
        hsd[q + 'self_attn.out_proj.bias'] = sd[p + 'attn.out_proj.bias']
        hsd[q + 'layer_norm1.weight'] = sd[p + 'ln_1.weight']
        hsd[q + 'layer_norm1.bias'] = sd[p + 'ln_1.bias']
        hsd[q + 'layer_norm2.weight'] = sd[p + 'ln_2.weight']
        hsd[q + 'layer_norm2.bias'] = sd[p + 'ln_2.bias']
        hsd[q + 'mlp.fc1.weight'] = sd[p + 'mlp.c_fc.weight']
        hsd[q + 'mlp.fc1.bias'] = sd[p + 'mlp.c_fc.bias']
        hsd[q + 'mlp.fc2.weight'] = sd[p + 'mlp.c_proj.weight']
        hsd[q + 'mlp.fc2.bias'] = sd[p + 'mlp.c_proj.bias']
    missing, unexpected = hf.load_state_dict(hsd, strict=False)
    assert not unexpected and all('position_ids' in m for m in missing), (missing, unexpected)
    x = synthetic_images(3, seed=2)
    with torch.no_grad():
        ref = hf(pixel_values=x).image_embeds
    out = encode_image_ref(sd, ViTConfig(**TINY), x)
    torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


def test_dual_stream_matches_reference_hooks():
    z = np.load(GOLD / 'hooks_tiny.npz')
    arch = json.loads(str(z['arch']))
    sd = synthetic_state_dict(**arch, seed=int(z['seed']))
    x = synthetic_images(3, seed=int(z['image_seed']))
    cfg_plain = ViTConfig(**arch)
    torch.testing.assert_close(encode_image_ref(sd, cfg_plain, x), torch.from_numpy(z['plain']),
                               rtol=1e-4, atol=1e-4)
    sd2 = dict(sd)
    sd2['visual.positional_embedding'] = torch.from_numpy(z['pos'])
    cfg = ViTConfig(**arch, stride=16, padding=15)
    assert cfg.grid == 14 and cfg.tokens == 197
    masks = torch.from_numpy(z['masks'])
    out = encode_objects_ref(sd2, cfg, x, masks)
    torch.testing.assert_close(out, torch.from_numpy(z['objects']), rtol=1e-4, atol=1e-4)
    out0 = encode_objects_ref(sd2, cfg, x, torch.zeros_like(masks))
    torch.testing.assert_close(out0, torch.from_numpy(z['objects_all_fg']), rtol=1e-4, atol=1e-4)
    # the mask path is live: masked != all-foreground for the masked crops
    assert (out - out0).abs().max() > 1e-3


def test_positional_embedding_interpolation_matches_fixture():
    """Product-side surgery (oadp_amd.oake.objects.Validator._build_model) produces the same
    interpolated positional embedding the golden run used."""
    from oadp_amd.clip.model import VisionTransformer
    z = np.load(GOLD / 'hooks_tiny.npz')
    arch = json.loads(str(z['arch']))
    sd = synthetic_state_dict(**arch, seed=int(z['seed']))

    class _V:
        positional_embedding = sd['visual.positional_embedding']
    pos = VisionTransformer.interpolate_positional_embedding(_V, (14, 14))
    assert pos.shape == (197, arch['width'])
    torch.testing.assert_close(pos, torch.from_numpy(z['pos']), rtol=1e-6, atol=1e-6)
    assert torch.equal(pos[0], sd['visual.positional_embedding'][0])  # CLS row untouched


def test_text_oracle_matches_huggingface_clip():
    """oracle/text_ref.py (encode_text restatement) == HuggingFace CLIPTextModelWithProjection on shared
    random weights (quick_gelu, causal mask, EOT = argmax of the token ids)."""
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection

    from oadp_amd.weights import synthetic_text_state_dict, synthetic_tokens
    from oracle.text_ref import TextConfig, encode_text_ref
    arch = dict(context=20, vocab=300, width=128, layers=2, heads=2, mlp_dim=256, embed_dim=64)
    W = arch['width']
    sd = synthetic_text_state_dict(**arch)
    hc = CLIPTextConfig(vocab_size=300, hidden_size=W, intermediate_size=256, projection_dim=64,
                        num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=20,
                        hidden_act='quick_gelu', eos_token_id=2, bos_token_id=0, pad_token_id=1)
    m = CLIPTextModelWithProjection(hc).eval()
    hs = m.state_dict()

    def put(k, v):
        assert hs[k].shape == v.shape, (k, hs[k].shape, v.shape)
        hs[k] = v.clone()

    put('text_model.embeddings.token_embedding.weight', sd['token_embedding.weight'])
    put('text_model.embeddings.position_embedding.weight', sd['positional_embedding'])
    for i in range(2):
        p, q = f'transformer.resblocks.{i}.', f'text_model.encoder.layers.{i}.'
        w, b = sd[p + 'attn.in_proj_weight'], sd[p + 'attn.in_proj_bias']
        for j, nm in enumerate(('q_proj', 'k_proj', 'v_proj')):
            put(q + f'self_attn.{nm}.weight', w[j * W:(j + 1) * W])
            put(q + f'self_attn.{nm}.bias', b[j * W:(j + 1) * W])
        put(q + 'self_attn.out_proj.weight', sd[p + 'attn.out_proj.weight'])
        put(q + 'self_attn.out_proj.bias', sd[p + 'attn.out_proj.bias'])
        for a, c in (('layer_norm1', 'ln_1'), ('layer_norm2', 'ln_2')):
            put(q + a + '.weight', sd[p + c + '.weight'])
            put(q + a + '.bias', sd[p + c + '.bias'])
        put(q + 'mlp.fc1.weight', sd[p + 'mlp.c_fc.weight'])
        put(q + 'mlp.fc1.bias', sd[p + 'mlp.c_fc.bias'])
        put(q + 'mlp.fc2.weight', sd[p + 'mlp.c_proj.weight'])
        put(q + 'mlp.fc2.bias', sd[p + 'mlp.c_proj.bias'])
    put('text_model.final_layer_norm.weight', sd['ln_final.weight'])
    put('text_model.final_layer_norm.bias', sd['ln_final.bias'])
    put('text_projection.weight', sd['text_projection'].t().contiguous())
    m.load_state_dict(hs)
    tok = synthetic_tokens(6, 20, 300)
    with torch.no_grad():
        ref = m(input_ids=tok.long()).text_embeds
    torch.testing.assert_close(encode_text_ref(sd, TextConfig(**arch), tok), ref, rtol=1e-4, atol=1e-4)
