"""Deterministic synthetic CLIP-ViT weights and inputs (no checkpoints, no network).

A repo-owned counter-based generator (splitmix64 -> uniform) keyed by tensor NAME, so any box
regenerates bit-identical fp32 tensors without relying on ``torch.manual_seed`` stability.
Keys/shapes are the OpenAI-CLIP ``visual.*`` state_dict (SURVEY.md §7 hard part 1), so a real
``ViT-B-32.pt`` state_dict drops into the same loader.
"""
from __future__ import annotations

import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform(name: str, shape: tuple[int, ...], seed: int = 0) -> np.ndarray:
    """U[0,1) float64 array, a pure function of (name, seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over='ignore'):
        z = base + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def _sym(name: str, shape, bound: float, seed: int) -> torch.Tensor:
    return torch.from_numpy(((uniform(name, tuple(shape), seed) * 2 - 1) * bound).astype(np.float32))


def normal(name: str, shape, seed: int = 0) -> torch.Tensor:
    """Approximately N(0,1) (sum of 4 uniforms, variance-matched) — for synthetic images."""
    u = sum(uniform(f'{name}#{k}', tuple(shape), seed) for k in range(4))
    return torch.from_numpy(((u - 2.0) * np.sqrt(3.0)).astype(np.float32))


def synthetic_state_dict(*, image_size=224, patch_size=32, width=768, layers=12, heads=12,
                         mlp_dim=3072, embed_dim=512, seed: int = 1) -> dict[str, torch.Tensor]:
    """Random-init vision-tower state_dict (fp32, CPU) of the given architecture."""
    del heads
    grid = image_size // patch_size
    sd: dict[str, torch.Tensor] = {}
    s = width ** -0.5

    def ln(prefix: str) -> None:
        sd[prefix + '.weight'] = 1.0 + _sym(prefix + '.weight', (width,), 0.1, seed)
        sd[prefix + '.bias'] = _sym(prefix + '.bias', (width,), 0.1, seed)

    k = 3 * patch_size * patch_size
    sd['visual.conv1.weight'] = _sym('visual.conv1.weight', (width, 3, patch_size, patch_size),
                                     k ** -0.5, seed)
    sd['visual.class_embedding'] = _sym('visual.class_embedding', (width,), s * 1.7, seed)
    sd['visual.positional_embedding'] = _sym('visual.positional_embedding',
                                             (grid * grid + 1, width), s * 1.7, seed)
    ln('visual.ln_pre')
    for i in range(layers):
        p = f'visual.transformer.resblocks.{i}.'
        ln(p + 'ln_1')
        ln(p + 'ln_2')
        sd[p + 'attn.in_proj_weight'] = _sym(p + 'attn.in_proj_weight', (3 * width, width), width ** -0.5, seed)
        sd[p + 'attn.in_proj_bias'] = _sym(p + 'attn.in_proj_bias', (3 * width,), 0.05, seed)
        sd[p + 'attn.out_proj.weight'] = _sym(p + 'attn.out_proj.weight', (width, width), width ** -0.5, seed)
        sd[p + 'attn.out_proj.bias'] = _sym(p + 'attn.out_proj.bias', (width,), 0.05, seed)
        sd[p + 'mlp.c_fc.weight'] = _sym(p + 'mlp.c_fc.weight', (mlp_dim, width), width ** -0.5, seed)
        sd[p + 'mlp.c_fc.bias'] = _sym(p + 'mlp.c_fc.bias', (mlp_dim,), 0.05, seed)
        sd[p + 'mlp.c_proj.weight'] = _sym(p + 'mlp.c_proj.weight', (width, mlp_dim), mlp_dim ** -0.5, seed)
        sd[p + 'mlp.c_proj.bias'] = _sym(p + 'mlp.c_proj.bias', (width,), 0.05, seed)
    ln('visual.ln_post')
    sd['visual.proj'] = _sym('visual.proj', (width, embed_dim), s * 1.7, seed)
    return sd


def synthetic_text_state_dict(*, context=77, vocab=49408, width=512, layers=12, heads=8, mlp_dim=2048,
                              embed_dim=512, seed: int = 2) -> dict[str, torch.Tensor]:
    """Random-init text-tower state_dict (fp32, CPU): the CLIP names without the ``visual.`` prefix."""
    del heads
    sd: dict[str, torch.Tensor] = {}

    def ln(prefix: str) -> None:
        sd[prefix + '.weight'] = 1.0 + _sym(prefix + '.weight', (width,), 0.1, seed)
        sd[prefix + '.bias'] = _sym(prefix + '.bias', (width,), 0.1, seed)

    sd['token_embedding.weight'] = _sym('token_embedding.weight', (vocab, width), 0.05, seed)
    sd['positional_embedding'] = _sym('positional_embedding', (context, width), 0.03, seed)
    for i in range(layers):
        p = f'transformer.resblocks.{i}.'
        ln(p + 'ln_1')
        ln(p + 'ln_2')
        sd[p + 'attn.in_proj_weight'] = _sym(p + 'attn.in_proj_weight', (3 * width, width), width ** -0.5, seed)
        sd[p + 'attn.in_proj_bias'] = _sym(p + 'attn.in_proj_bias', (3 * width,), 0.05, seed)
        sd[p + 'attn.out_proj.weight'] = _sym(p + 'attn.out_proj.weight', (width, width), width ** -0.5, seed)
        sd[p + 'attn.out_proj.bias'] = _sym(p + 'attn.out_proj.bias', (width,), 0.05, seed)
        sd[p + 'mlp.c_fc.weight'] = _sym(p + 'mlp.c_fc.weight', (mlp_dim, width), width ** -0.5, seed)
        sd[p + 'mlp.c_fc.bias'] = _sym(p + 'mlp.c_fc.bias', (mlp_dim,), 0.05, seed)
        sd[p + 'mlp.c_proj.weight'] = _sym(p + 'mlp.c_proj.weight', (width, mlp_dim), mlp_dim ** -0.5, seed)
        sd[p + 'mlp.c_proj.bias'] = _sym(p + 'mlp.c_proj.bias', (width,), 0.05, seed)
    ln('ln_final')
    sd['text_projection'] = _sym('text_projection', (width, embed_dim), width ** -0.5, seed)
    return sd


def synthetic_tokens(n: int, length: int = 77, vocab: int = 49408, seed: int = 0) -> torch.Tensor:
    """[n, length] int32 token rows shaped like clip.tokenize output: SOT, words, EOT (the highest id,
    at a different position per row), zero padding."""
    g = torch.Generator().manual_seed(1000 + seed)
    out = torch.zeros(n, length, dtype=torch.int32)
    for i in range(n):
        k = int(torch.randint(1, max(2, length - 1), (1,), generator=g))
        out[i, 0] = vocab - 2
        out[i, 1:k] = torch.randint(1, vocab - 2, (k - 1,), generator=g, dtype=torch.int32)
        out[i, k] = vocab - 1
    return out


def synthetic_images(n: int, image_size: int = 224, seed: int = 0) -> torch.Tensor:
    """[n,3,S,S] fp32 ~N(0,1): the shape/statistics of CLIP-normalised crops."""
    return normal('images', (n, 3, image_size, image_size), seed)
