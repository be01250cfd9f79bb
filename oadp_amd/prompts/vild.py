"""Category-name text embeddings with the ViLD prompt ensemble — mirror of oadp/prompts/vild.py.

Reference flow (vild.py:54-72): for each of the 74 templates, format every category name, tokenize
(``clip.adaptively_tokenize``), ``model.encode_text``, L2-normalise; average the 74 normalised
embeddings per category; save ``dict(embeddings=[K, 512], names=[K])`` to data/prompts/vild.pth.
Here ``encode_text`` is ``oake_encode_text`` (csrc: the vision tower's kernels with a causal mask).

Not shipped with this repo: the category lists (``oadp.base.coco / lvis``: pass ``categories``) and the
BPE vocabulary of CLIP's tokenizer (pass ``encode``: text -> list of token ids without SOT/EOT).
"""
from __future__ import annotations

import pathlib
from typing import Callable, Iterable, Sequence

import torch

SOT, EOT, CONTEXT = 49406, 49407, 77  # clip/simple_tokenizer.py: <|startoftext|>, <|endoftext|>


def templates() -> list[str]:
    """The 74 ViLD templates in the reference's order (oadp/prompts/vild.py:9-51): they are the product
    of a few word choices — {This is, There is} x {a, the, one} x {-, small, medium, large} x
    {-, in the scene / photo / picture} plus the "a photo of" family — generated here."""
    sizes = ['', 'small ', 'medium ', 'large ']
    arts = ['a', 'the', 'one']
    out = ['This is a {}', 'There is a {}']
    out += [f'a photo of a {s}{{}} in the scene' for s in sizes]
    out += [f'a photo of a {s}{{}}' for s in sizes]
    out += [f'This is a photo of a {s}{{}}' for s in sizes]
    out += [f'There is {a} {{}} in the scene' for a in arts]
    out += [f'This is {a} {{}} in the scene' for a in arts]
    out += [f'This is one {s}{{}} in the scene' for s in sizes[1:]]
    out += [f'There is a {s}{{}} in the scene' for s in sizes[1:]]
    for head, place in (('There is', 'photo'), ('There is', 'picture'), ('This is', 'photo'),
                        ('This is', 'picture')):
        out += [f'{head} {a} {s}{{}} in the {place}' for s in sizes for a in arts]
    return out


def adaptively_tokenize(texts: Iterable[str], encode: Callable[[str], Sequence[int]], *,
                        context: int = CONTEXT, sot: int = SOT, eot: int = EOT) -> torch.Tensor:
    """``clip.tokenize`` with the context trimmed to the longest text of the batch (the fork's
    ``adaptively_tokenize``; legal for a causal text tower): [n, L] int32, rows = SOT ids EOT 0..."""
    rows = [[sot, *encode(t), eot] for t in texts]
    length = max(len(r) for r in rows)
    if length > context:
        raise ValueError(f'a prompt needs {length} tokens, context is {context}')
    out = torch.zeros(len(rows), length, dtype=torch.int32)
    for i, r in enumerate(rows):
        out[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
    return out


def embed(model, categories: Sequence[str], encode: Callable[[str], Sequence[int]], *,
          device: torch.device | str = 'cuda', prompts: Sequence[str] | None = None,
          dtype: torch.dtype = torch.float32, **tok) -> dict:
    """The reference's loop (oadp/prompts/vild.py:60-71): per template, encode_text -> F.normalize; mean over
    the templates.  ``dtype``: element type of the saved ``embeddings``.  The reference saves whatever
    ``model.encode_text`` returns — fp16 when its CLIP runs on a GPU, fp32 on the CPU; the template mean is
    accumulated in fp32 here either way (default fp32 out; pass ``torch.float16`` for the GPU reference's file)."""
    total = None
    prompts = list(prompts) if prompts is not None else templates()
    for prompt in prompts:
        tokens = adaptively_tokenize(map(prompt.format, categories), encode, **tok)
        e = model.encode_text(tokens.to(device), normalize=True, out_dtype=torch.float32)  # F.normalize fused
        total = e if total is None else total + e
    return dict(embeddings=(total / len(prompts)).to(dtype).cpu(), names=list(categories))


def main(categories: Sequence[str], encode: Callable[[str], Sequence[int]], *, model=None,
         output: str = 'data/prompts/vild.pth', **kwargs) -> dict:
    if model is None:
        from .. import clip
        model, _ = clip.load_default()
    state = embed(model, sorted(set(categories)), encode, **kwargs)
    path = pathlib.Path(output)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(state, path)
    return state
