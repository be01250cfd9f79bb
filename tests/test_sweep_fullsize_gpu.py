"""BASELINE.json configs[4] AT SIZE, with values: rank 0's whole share of the 8-rank, 118 k-image OAKE sweep
through the product's three entry points, and what it wrote checked against the oracle.

  globals + blocks   the full share: 14 750 synthetic JPEGs (+ 5 val) of the 118 000 listed, OAKE_SHARD=0/8, every
                     shipped default (device decode, flush sizes, `.pth` writer): 2 x 14 755 files;
  objects            the same sharding over a 16 000-image listing: 2 000 images x 300 proposals = 600 k crops
                     (objects mode runs at ~90 images/s per GPU; the full share is 164 s of the same loop,
                     profiles/r04/sweep_shard_rank0of8_118k_final.log)
[REF README.md:197-229; oadp/oake/base.py:85-152; oadp/dp/datasets.py:171-214].

Unlike tests/test_sweep_gpu.py (counts and read-back of a 96-image tree) this test checks VALUES: for a seeded
sample of 32 written files per tree, the file's rows against the CPU oracle run from the JPEG file itself —
Pillow decode -> oracle/crops_ref (pyramid / expand / masks / PIL crops + CLIP transform) -> oracle/vit_ref (fp32
encoder) -> L2-normalise:

  * bboxes / objectness: bit-exact for every row of every sampled file (the index math of the path);
  * embeddings: BASELINE.json north_star tolerance — fp16 rtol 1e-3 / atol 1e-3 (+ the fp16 storage rounding of
    the file, <= 2.5e-4 on unit-norm features), cosine >= 0.999 — on every row (globals), block 0 + 3 seeded rows
    per file (blocks), 4 seeded rows per file (objects; the oracle's dual-stream encoder runs ~10 crops/s).
"""
import json
import os
import pathlib
import shutil
import subprocess
import sys
import time

import numpy as np
import PIL.Image
import pytest
import torch

from oadp_amd.clip.model import VisionTransformer
from oadp_amd.weights import synthetic_state_dict
from oracle import crops_ref
from oracle.vit_ref import ViTConfig, encode_image_ref, encode_objects_ref, l2_normalize

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tools'))
N_FILES = 32


def _sweep(root: pathlib.Path, total: int, modes: str) -> dict:
    r = subprocess.run([sys.executable, str(ROOT / 'tools' / 'sweep_shard.py'), '--total', str(total), '--world', '8',
                        '--rank', '0', '--root', str(root), '--modes', modes, '--sample', '32'],
                       capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert lines[-1]['ok'] and lines[-1]['shard'] == '0/8'
    return {d['mode']: d for d in lines[1:-1]}


def _close(got: torch.Tensor, ref: torch.Tensor, what: str) -> None:
    got, ref = got.float(), ref.float()
    cos = torch.nn.functional.cosine_similarity(got, ref, dim=1)
    print(f'{what}: rows={got.shape[0]} max|err|={(got - ref).abs().max().item():.3e} min cos={cos.min().item():.6f}')
    assert cos.min().item() >= 0.999, what
    torch.testing.assert_close(got, ref, rtol=1e-3, atol=1.25e-3)


# Which files and which rows the oracle checks differs from run to run (a fixed choice could
# hide a defect that hits only, say, proposals clipped at the right image border for ever): OAKE_TEST_SEED names the base
# seed, default = derived from the clock; it is printed, so a failing draw can be replayed.
_SEED = int(os.environ['OAKE_TEST_SEED']) if os.environ.get('OAKE_TEST_SEED') else int(time.time()) % 1_000_000
print(f'[test_sweep_fullsize_gpu] sample seed OAKE_TEST_SEED={_SEED}', flush=True)


def _pick(owned: list[int], seed: int) -> list[int]:
    rng = np.random.default_rng(_SEED * 100 + seed)
    return sorted(int(i) for i in rng.choice(owned, size=N_FILES, replace=False))


@pytest.fixture(scope='module')
def vit_b32():
    return synthetic_state_dict()  # what OAKE_SYNTHETIC_WEIGHTS=1 loads in the entry points


def test_rank0_share_globals_blocks_at_size(cuda, tmp_path, vit_b32):
    root = tmp_path / 'coco'
    try:
        modes = _sweep(root, 118_000, 'globals,blocks')
        for m in ('globals', 'blocks'):  # every owned image exactly once: 118 000 / 8 train files + 5 val
            assert modes[m]['rc'] == 0 and modes[m]['files_train'] == 14_750 and modes[m]['files_val'] == 5, modes[m]
        owned = list(range(0, 118_000, 8))
        cfg = ViTConfig()
        # ---- globals: every sampled file is one row
        ids = _pick(owned, seed=11)
        crops, got = [], []
        for i in ids:
            pil = PIL.Image.open(root / 'train2017' / f'{i:012d}.jpg').convert('RGB')
            # globals mode's transform is clip.load_default(True) [REF oadp/oake/globals.py:47]: with the shipped
            # fork setting (oadp_amd/clip/settings.py, load_default_true='squash') Resize((224, 224)) without a crop
            crops.append(torch.from_numpy(crops_ref.preprocess_ref(pil.resize((224, 224), PIL.Image.BICUBIC))))
            t = torch.load(root / 'oake' / 'globals' / 'train2017' / f'{i:012d}.pth', 'cpu')
            assert t.dtype == torch.float16 and t.shape == (512,)
            got.append(t)
        _close(torch.stack(got), l2_normalize(encode_image_ref(vit_b32, cfg, torch.stack(crops))), 'globals')
        # ---- blocks: bboxes of every row bit-exact; block 0 + 3 seeded rows per file against the oracle
        ids = _pick(owned, seed=12)
        rng = np.random.default_rng(_SEED * 100 + 13)
        crops, got = [], []
        for i in ids:
            pil = PIL.Image.open(root / 'train2017' / f'{i:012d}.jpg').convert('RGB')
            w, h = pil.size
            d = torch.load(root / 'oake' / 'blocks' / 'train2017' / f'{i:012d}.pth', 'cpu')
            assert d['embeddings'].dtype == torch.float16 and d['bboxes'].dtype == torch.float16
            want = torch.from_numpy(crops_ref.all_block_bboxes(w, h)).half()
            assert d['embeddings'].shape == (want.shape[0], 512) and want.shape[0] == 27
            assert torch.equal(d['bboxes'], want), i
            tiles = list(crops_ref.partitions(w, h))  # (level w, level h, scale, x, y), reference order
            rows = [0] + sorted(int(r) for r in rng.choice(np.arange(1, 27), size=3, replace=False))
            levels = {}
            for r in rows:
                if r == 0:
                    crops.append(torch.from_numpy(crops_ref.preprocess_ref(pil)))
                else:
                    lw, lh, _, x, y = tiles[r - 1]
                    if (lw, lh) not in levels:  # the pyramid is resized level by level from the previous level
                        level = pil
                        for pw, ph, _ in crops_ref.pyramid_sizes(w, h):
                            if level.size != (pw, ph):
                                level = level.resize((pw, ph))
                            levels[(pw, ph)] = level
                    crops.append(torch.from_numpy(crops_ref.preprocess_ref(levels[(lw, lh)].crop((x, y, x + 224, y + 224)))))
                got.append(d['embeddings'][r])
        _close(torch.stack(got), l2_normalize(encode_image_ref(vit_b32, cfg, torch.stack(crops))), 'blocks')
    finally:
        shutil.rmtree(root, ignore_errors=True)


def test_rank0_share_objects_at_size(cuda, tmp_path, vit_b32):
    import sweep_shard  # tools/: the proposal generator of the synthetic tree (seeded by image id)
    root = tmp_path / 'coco'
    try:
        modes = _sweep(root, 16_000, 'objects')
        assert modes['objects']['rc'] == 0 and modes['objects']['files_train'] == 2_000, modes['objects']
        owned = list(range(0, 16_000, 8))
        ids = _pick(owned, seed=21)
        rng = np.random.default_rng(_SEED * 100 + 22)
        sd = dict(vit_b32)  # the reference's surgery: positional embedding on the 14 x 14 grid, stride 16, padding 15
        holder = type('P', (), {'positional_embedding': sd['visual.positional_embedding']})()
        sd['visual.positional_embedding'] = VisionTransformer.interpolate_positional_embedding(holder, (14, 14))
        cfg = ViTConfig(stride=16, padding=15)
        crops, masks, got = [], [], []
        for i in ids:
            pil = PIL.Image.open(root / 'train2017' / f'{i:012d}.jpg').convert('RGB')
            w, h = pil.size
            prop = sweep_shard._proposals(i, w, h)
            keep = crops_ref.keep_min_wh(prop[:, :4])
            d = torch.load(root / 'oake' / 'objects' / 'train2017' / f'{i:012d}.pth', 'cpu')
            n = int(keep.sum())
            assert n > 200 and d['embeddings'].shape == (n, 512) and d['embeddings'].dtype == torch.float16
            # index math: the un-expanded, min_wh-filtered proposals and their scores, bit for bit
            assert torch.equal(d['bboxes'], torch.from_numpy(prop[keep, :4]).half()), i
            assert torch.equal(d['objectness'], torch.from_numpy(prop[keep, 4:]).half()), i
            boxes = crops_ref.expand_adaptive(prop[keep, :4], (w, h))
            for r in sorted(int(r) for r in rng.choice(n, size=4, replace=False)):
                b = boxes[r]
                crops.append(torch.from_numpy(crops_ref.preprocess_ref(pil.crop(tuple(float(c) for c in b)))))
                fg = prop[keep][r, :4] - np.concatenate([b[:2], b[:2]])
                masks.append(torch.from_numpy(crops_ref.object_mask(fg, b)).float()[None])
                got.append(d['embeddings'][r])
        ref = l2_normalize(encode_objects_ref(sd, cfg, torch.stack(crops), torch.stack(masks)))
        _close(torch.stack(got), ref, 'objects')
    finally:
        shutil.rmtree(root, ignore_errors=True)
