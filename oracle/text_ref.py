"""TEST INFRASTRUCTURE ONLY — fp32 torch-CPU restatement of CLIP's text tower (``model.encode_text``,
called by oadp/prompts/vild.py:62-66).  The arithmetic lives in the un-vendored LutingWang/CLIP fork
(see oracle/__init__.py): parity against the fork is unpinned; this restatement is pinned against
HuggingFace ``CLIPTextModelWithProjection`` on shared random weights (tests/test_oracle_vit.py).

Structure (OpenAI clip/model.py encode_text): x = token_embedding(text) + positional_embedding[:L];
12 pre-LN residual blocks with a causal additive mask; ln_final; the row at text.argmax(-1) (the EOT
token has the highest id) @ text_projection.
"""
from __future__ import annotations

import dataclasses
from typing import Mapping

import torch
import torch.nn.functional as F


@dataclasses.dataclass
class TextConfig:
    context: int = 77
    vocab: int = 49408
    width: int = 512
    layers: int = 12
    heads: int = 8
    mlp_dim: int = 2048
    embed_dim: int = 512


def encode_text_ref(sd: Mapping[str, torch.Tensor], cfg: TextConfig, tokens: torch.Tensor) -> torch.Tensor:
    """tokens int [n, L] (L <= context) -> [n, embed_dim] fp32, un-normalised."""
    tokens = tokens.long()
    n, L = tokens.shape
    C, H = cfg.width, cfg.heads
    x = sd['token_embedding.weight'].float()[tokens] + sd['positional_embedding'].float()[:L]
    mask = torch.full((L, L), float('-inf')).triu_(1)  # clip build_attention_mask
    for i in range(cfg.layers):
        p = f'transformer.resblocks.{i}.'
        y = F.layer_norm(x, (C,), sd[p + 'ln_1.weight'].float(), sd[p + 'ln_1.bias'].float(), 1e-5)
        qkv = y @ sd[p + 'attn.in_proj_weight'].float().t() + sd[p + 'attn.in_proj_bias'].float()
        q, k, v = (t.reshape(n, L, H, C // H).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        a = torch.softmax(q @ k.transpose(-1, -2) * (C // H) ** -0.5 + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(n, L, C)
        x = x + a @ sd[p + 'attn.out_proj.weight'].float().t() + sd[p + 'attn.out_proj.bias'].float()
        y = F.layer_norm(x, (C,), sd[p + 'ln_2.weight'].float(), sd[p + 'ln_2.bias'].float(), 1e-5)
        h = y @ sd[p + 'mlp.c_fc.weight'].float().t() + sd[p + 'mlp.c_fc.bias'].float()
        h = h * torch.sigmoid(1.702 * h)
        x = x + h @ sd[p + 'mlp.c_proj.weight'].float().t() + sd[p + 'mlp.c_proj.bias'].float()
    x = F.layer_norm(x, (C,), sd['ln_final.weight'].float(), sd['ln_final.bias'].float(), 1e-5)
    eot = tokens.argmax(dim=-1)
    return x[torch.arange(n), eot] @ sd['text_projection'].float()
