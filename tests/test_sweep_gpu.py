"""BASELINE.json configs[4] in miniature, as a test: one rank's share of an 8-rank OAKE sweep — a synthetic COCO-like
tree (JPEG files for the rank's DistributedSampler shard, proposals for them), the three entry points
`python -m oadp_amd.oake.{globals,blocks,objects}` with OAKE_SHARD=r/8 and every shipped default (device decode,
flush sizes, `.pth` writer), then the detector side's `LoadCLIPFeatures` over the files as written
[REF README.md:197-207; oadp/oake/base.py:85-126; oadp/dp/datasets.py:171-214].  tools/sweep_shard.py is the same
thing at 118 k images (profiles/r0*/sweep_shard_*.log); here: 96 listed, 12 owned by rank 5."""
import json
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.mark.parametrize('writer', ['pth', 'pack'])
def test_one_rank_of_an_eight_rank_sweep(tmp_path, writer):
    r = subprocess.run([sys.executable, str(ROOT / 'tools' / 'sweep_shard.py'), '--total', '96', '--world', '8', '--rank',
                        '5', '--root', str(tmp_path), '--sample', '12', '--writer', writer],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith('{')]
    tree, modes, last = lines[0], {d['mode']: d for d in lines[1:-1]}, lines[-1]
    assert tree['train'] == dict(listed=96, owned=12) and tree['val']['listed'] == 40
    assert set(modes) == {'globals', 'blocks', 'objects'}
    for d in modes.values():  # every owned image exactly once, in each of the three trees
        assert d['rc'] == 0 and d['files_train'] == 12 and d['files_val'] == 5, d
    assert last['ok'] and last['samples'] == 12 and last['shard'] == '5/8'
    if writer == 'pth':
        names = sorted(p.name for p in (tmp_path / 'oake' / 'objects' / 'train2017').glob('*.pth'))
        assert names == [f'{i:012d}.pth' for i in range(5, 96, 8)]  # rank 5 of 8: ids 5, 13, ..., 93
