"""Category-name text embeddings with the ViLD prompt ensemble — mirror of oadp/prompts/vild.py.

Reference flow (vild.py:54-72): for each of the 74 templates, format every category name, tokenize
(``clip.adaptively_tokenize``), ``model.encode_text``, L2-normalise; average the 74 normalised
embeddings per category; save ``dict(embeddings=[K, 512], names=[K])`` to data/prompts/vild.pth.
Here ``encode_text`` is ``oake_encode_text`` (csrc: the vision tower's kernels with a causal mask).

``python -m oadp_amd.prompts.vild`` is the reference's entry point [REF oadp/prompts/vild.py:54-76].  Two inputs are
assets this repo does not carry and takes as paths: the category NAMES — read from the datasets' own annotation
files (``categories[].name`` of the OV-COCO / LVIS v1 JSONs the reference trains on: the same strings as
``oadp.base.coco.all_ + lvis.all_``) or from a one-name-per-line text file — and the BPE vocabulary of CLIP's
tokenizer (``bpe_simple_vocab_16e6.txt.gz``; ``oadp_amd/prompts/bpe.py`` restates the algorithm).  As a library:
pass ``categories`` and ``encode`` (text -> token ids without SOT / EOT).
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


def read_categories(paths: Sequence[str]) -> list[str]:
    """Category names of COCO / LVIS-format annotation files (``categories[].name``) and / or text files with one
    name per line; the union, as the reference's ``sorted(set(coco.all_ + lvis.all_))``."""
    import json
    names: set[str] = set()
    for p in paths:
        text = pathlib.Path(p).read_text()
        if p.endswith('.json'):
            names.update(c['name'] for c in json.loads(text)['categories'])
        else:
            names.update(line.strip() for line in text.splitlines() if line.strip())
    return sorted(names)


def cli(argv: Sequence[str] | None = None) -> None:
    import argparse
    from .bpe import Tokenizer
    ap = argparse.ArgumentParser(description='ViLD prompt-ensemble text embeddings of the category names '
                                             '(data/prompts/vild.pth of the reference)')
    ap.add_argument('--categories-from', nargs='+',
                    default=['data/coco/annotations/instances_val2017.65.min.json',
                             'data/lvis_v1/annotations/lvis_v1_val.json'],
                    help='annotation JSONs (categories[].name) and / or one-name-per-line text files')
    ap.add_argument('--bpe', default='pretrained/clip/bpe_simple_vocab_16e6.txt.gz', help="CLIP's BPE vocabulary file")
    ap.add_argument('--output', default='data/prompts/vild.pth')
    ap.add_argument('--dtype', choices=['float32', 'float16'], default='float32',
                    help='float16: the file the reference writes when its CLIP runs on a GPU')
    a = ap.parse_args(argv)
    state = main(read_categories(a.categories_from), Tokenizer(a.bpe).encode, output=a.output,
                 dtype=getattr(torch, a.dtype))
    print(f'{a.output}: {len(state["names"])} categories x {state["embeddings"].shape[1]}')


if __name__ == '__main__':
    cli()
