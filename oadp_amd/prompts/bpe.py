"""CLIP's byte-level BPE tokenizer, restated from the published algorithm (OpenAI CLIP ``simple_tokenizer.py``; the
reference reaches it through ``clip.adaptively_tokenize`` [REF oadp/prompts/vild.py:62-64]).  The vocabulary file —
``bpe_simple_vocab_16e6.txt.gz``, 262 145 lines — is an asset of the CLIP package and is not shipped here: pass its path.

    text -> lower-case, whitespace collapsed -> regex pieces -> every piece: UTF-8 bytes mapped to printable
    code points, last one + '</w>', greedily merged by merge rank -> ids: 256 byte symbols, 256 byte symbols + '</w>',
    the 48 894 merges, <|startoftext|> = 49406, <|endoftext|> = 49407.

(``ftfy.fix_text`` of the original's ``basic_clean`` is not applied — ftfy is not a dependency here; it is the identity
on the plain-ASCII category names and templates this module is used for.)"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache

import regex

N_MERGES = 49152 - 256 - 2
PATTERN = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                        regex.IGNORECASE)


@lru_cache()
def bytes_to_unicode() -> dict[int, str]:
    """Every byte as a printable code point: the printable Latin-1 ranges map to themselves, the other 68 bytes to
    256, 257, ... in byte order."""
    keep = list(range(ord('!'), ord('~') + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    table, n = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + n)
            n += 1
    return table


class Tokenizer:

    def __init__(self, vocab_path: str | os.PathLike) -> None:
        opener = gzip.open if str(vocab_path).endswith('.gz') else open
        with opener(vocab_path, 'rt', encoding='utf-8') as f:
            lines = f.read().split('\n')
        merges = [tuple(line.split()) for line in lines[1:N_MERGES + 1] if line.strip()]
        symbols = list(bytes_to_unicode().values())
        # byte symbols in CODE-POINT order of the table's values as the original builds them: first the kept bytes in
        # byte order, then the remapped ones
        keep = [c for b, c in sorted(bytes_to_unicode().items()) if ord(c) < 256]
        rest = [c for b, c in sorted(bytes_to_unicode().items()) if ord(c) >= 256]
        symbols = keep + rest
        vocab = symbols + [s + '</w>' for s in symbols] + [''.join(m) for m in merges]
        vocab += ['<|startoftext|>', '<|endoftext|>']
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.ranks = {m: i for i, m in enumerate(merges)}
        self._cache: dict[str, tuple[str, ...]] = {}

    def _bpe(self, piece: str) -> tuple[str, ...]:
        if piece in self._cache:
            return self._cache[piece]
        word = list(piece[:-1]) + [piece[-1] + '</w>']
        while len(word) > 1:
            pairs = [(self.ranks.get((a, b), float('inf')), i) for i, (a, b) in enumerate(zip(word, word[1:]))]
            rank, _ = min(pairs)
            if rank == float('inf'):
                break
            first, second = next(m for m, r in ((p, self.ranks.get(p)) for p in zip(word, word[1:])) if r == rank)
            merged, i = [], 0
            while i < len(word):  # every occurrence of the best pair, left to right
                if i + 1 < len(word) and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        self._cache[piece] = tuple(word)
        return self._cache[piece]

    def encode(self, text: str) -> list[int]:
        """Token ids of ``text`` without <|startoftext|> / <|endoftext|> (oadp_amd.prompts.vild adds them)."""
        text = ' '.join(html.unescape(html.unescape(text)).split()).strip().lower()
        b2u = bytes_to_unicode()
        ids: list[int] = []
        for piece in PATTERN.findall(text):
            mapped = ''.join(b2u[b] for b in piece.encode('utf-8'))
            ids.extend(self.encoder[t] for t in self._bpe(mapped))
        return ids
