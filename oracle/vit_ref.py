"""fp32 CPU restatement of the CLIP ViT image encoder used by OAKE.  Test infrastructure only.

Follows the structure of OpenAI ``clip/model.py`` (VisionTransformer / ResidualAttentionBlock),
constrained by every attribute the reference touches:
  conv1 / stride / padding / patch_size / grid      oadp/oake/objects.py:294-301
  positional_embedding                               oadp/oake/objects.py:292-296
  transformer.resblocks[i].{attn, ln_1, ln_2, mlp}   oadp/oake/objects.py:236-246, 308
  seq-first [L, N, C] activations                    oadp/oake/objects.py:222, 233-245
and of the reference's object-aware Hooks (oadp/oake/objects.py:198-266).

State-dict keys are the OpenAI checkpoint's (``visual.*``).  Everything is torch fp32 on CPU and
uses ``F.multi_head_attention_forward`` — the function behind ``nn.MultiheadAttention`` — so the
attention arithmetic (scaling of q, additive float mask, softmax) is PyTorch's own.
"""
from __future__ import annotations

import dataclasses

import torch
import torch.nn.functional as F


@dataclasses.dataclass(frozen=True)
class ViTConfig:
    image_size: int = 224
    patch_size: int = 32
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    embed_dim: int = 512
    stride: int = 32
    padding: int = 0

    @property
    def grid(self) -> int:
        return (self.image_size + 2 * self.padding - self.patch_size) // self.stride + 1

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def _blk(sd: dict, i: int, leaf: str) -> torch.Tensor:
    return sd[f'visual.transformer.resblocks.{i}.{leaf}'].float()


def _attn(sd: dict, i: int, heads: int, q, k, v, attn_mask=None) -> torch.Tensor:
    """nn.MultiheadAttention(width, heads)(q, k, v, need_weights=False, attn_mask=...)[0]"""
    e = q.shape[-1]
    out, _ = F.multi_head_attention_forward(
        q, k, v, e, heads,
        _blk(sd, i, 'attn.in_proj_weight'), _blk(sd, i, 'attn.in_proj_bias'),
        None, None, False, 0.0,
        _blk(sd, i, 'attn.out_proj.weight'), _blk(sd, i, 'attn.out_proj.bias'),
        training=False, need_weights=False, attn_mask=attn_mask,
    )
    return out


def _mlp(sd: dict, i: int, x: torch.Tensor) -> torch.Tensor:
    h = F.linear(x, _blk(sd, i, 'mlp.c_fc.weight'), _blk(sd, i, 'mlp.c_fc.bias'))
    return F.linear(quick_gelu(h), _blk(sd, i, 'mlp.c_proj.weight'), _blk(sd, i, 'mlp.c_proj.bias'))


def _ln(sd: dict, prefix: str, x: torch.Tensor) -> torch.Tensor:
    w = sd[prefix + '.weight'].float()
    return F.layer_norm(x, (w.shape[0],), w, sd[prefix + '.bias'].float(), 1e-5)


def _ln_blk(sd: dict, i: int, name: str, x: torch.Tensor) -> torch.Tensor:
    return _ln(sd, f'visual.transformer.resblocks.{i}.{name}', x)


def embed_tokens(sd: dict, cfg: ViTConfig, images: torch.Tensor) -> torch.Tensor:
    """conv1 -> flatten -> [cls; patches] + pos -> ln_pre -> seq-first [L, N, C]."""
    x = F.conv2d(images.float(), sd['visual.conv1.weight'].float(), None, cfg.stride, cfg.padding)
    n, c = x.shape[0], x.shape[1]
    x = x.reshape(n, c, -1).permute(0, 2, 1)  # [N, P2, C]
    cls = sd['visual.class_embedding'].float().expand(n, 1, c)
    x = torch.cat([cls, x], dim=1) + sd['visual.positional_embedding'].float()
    x = _ln(sd, 'visual.ln_pre', x)
    return x.permute(1, 0, 2)


def resblock(sd: dict, cfg: ViTConfig, i: int, x: torch.Tensor) -> torch.Tensor:
    h = _ln_blk(sd, i, 'ln_1', x)
    x = x + _attn(sd, i, cfg.heads, h, h, h)
    return x + _mlp(sd, i, _ln_blk(sd, i, 'ln_2', x))


def head(sd: dict, x_cls: torch.Tensor) -> torch.Tensor:
    return _ln(sd, 'visual.ln_post', x_cls) @ sd['visual.proj'].float()


@torch.no_grad()
def encode_image_ref(sd: dict, cfg: ViTConfig, images: torch.Tensor) -> torch.Tensor:
    """clip.model.CLIP.encode_image — [N,3,H,W] -> [N,embed] fp32 (not normalised)."""
    x = embed_tokens(sd, cfg, images)
    for i in range(cfg.layers):
        x = resblock(sd, cfg, i, x)
    return head(sd, x.permute(1, 0, 2)[:, 0, :])


@torch.no_grad()
def encode_objects_ref(sd: dict, cfg: ViTConfig, objects: torch.Tensor,
                       masks: torch.Tensor, return_layers: bool = False):
    """model.visual(objects, masks) under the reference's Hooks (objects.py:198-266).

    masks: [N,1,grid,grid], 1 = background.  Returns [N,embed] fp32 (not normalised).
    """
    n = objects.shape[0]
    # Hooks.visual_forward_pre: attn_mask = cat([mask 'b (h w)', zeros[b,1]]) * -100
    attn_mask = torch.cat([masks.float().reshape(n, -1), torch.zeros(n, 1)], dim=-1) * -100
    x = embed_tokens(sd, cfg, objects)
    y = x[[0]]  # Hooks.transformer_forward_pre
    # einops.repeat('b v -> (b h) 1 v'): batch-major, head-minor
    mask_bh = attn_mask.repeat_interleave(cfg.heads, dim=0).unsqueeze(1)
    ys = []
    for i in range(cfg.layers):
        # Hooks.residual_attention_block_forward_pre (runs before the block's own forward)
        xp = _ln_blk(sd, i, 'ln_1', torch.cat([x[1:], y]))
        y = y + _attn(sd, i, cfg.heads, xp[[-1]], xp, xp, attn_mask=mask_bh)
        y = y + _mlp(sd, i, _ln_blk(sd, i, 'ln_2', y))
        ys.append(y[0].clone())
        x = resblock(sd, cfg, i, x)
    out = head(sd, y[0])  # Hooks.transformer_forward returns y; then ln_post(x[:,0]) @ proj
    if return_layers:
        return out, ys
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """F.normalize(x) — dim 1, eps 1e-12."""
    return F.normalize(x.float())
