"""The N > 1 path's collective calls, executed ONCE on the real library (VERDICT r04 missing 1): every multi-rank run
that exists anywhere went over gloo (the CPU box, ranks sharing the 1-GPU box); `init_process_group('nccl')` — which
IS RCCL on ROCm — and the device-tensor collectives of the path had never run.  A world-size-1 group on the 1-GPU box
loads librccl, creates the communicator and runs those very calls:

  bench.py        OAKE_BENCH_FORCE_DIST=1: the counting all_reduce, the barriers around the timed region, the
                  max-over-ranks all_reduce of the elapsed time, the 32-byte counters all_gather
  validators      OAKE_FORCE_DIST=1 under the launcher's variables: init_process_group(pick_backend() == 'nccl'),
                  DistributedSampler over the group's rank / world, gather_counters' all_gather of 4 x f64 on the device
[REF oadp/oake/base.py:85-88,122-126; README.md:197-207]."""
import json
import os
import pathlib
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / 'tools'))


def _port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _env(**kw):
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='VERSION', **kw)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'OAKE_SHARD', 'OAKE_BENCH_BACKEND', 'OAKE_DIST_BACKEND'):
        if k not in kw:
            env.pop(k, None)
    return env


def test_bench_collectives_run_on_rccl_at_world_size_one(cuda):
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1',
                        '--no-profile', '--no-cpu-baseline', '--no-modes'], env=_env(OAKE_BENCH_FORCE_DIST='1'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['value'] > 1000 and line['config']['backend'] == 'nccl'
    assert 'ranks gathered: 1' in line['config']['sharding']
    assert 'torch.distributed.run' in line['config']['launcher']  # the multi-rank code path, not the single-process one
    print([ln for ln in (r.stdout + r.stderr).splitlines() if 'CCL version' in ln][:1])  # (NCCL_DEBUG=VERSION)


def test_validator_counters_gather_runs_on_rccl_at_world_size_one(cuda, tmp_path):
    import sweep_shard
    root = tmp_path / 'coco'
    sweep_shard.build_tree(root, 24, 1, 0, n_val=3)
    cfg = root / 'globals.py'
    cfg.write_text(sweep_shard.CONFIG.format(coco=str(root), mode='globals', batch=16, extra=''))
    env = _env(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', LOCAL_WORLD_SIZE='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_port()), OAKE_FORCE_DIST='1', OAKE_SYNTHETIC_WEIGHTS='1', TORCH_DISTRIBUTED_DEBUG='INFO')
    probe = ("import torch.distributed as td, atexit\n"
             "import oadp_amd.oake.base as b\n"
             "_g = b.gather_counters\n"
             "def g(c, d):\n"
             "    out = _g(c, d)\n"
             "    print('GATHER backend', td.get_backend(), 'world', td.get_world_size(), 'ranks', len(out),\n"
             "          'librccl mapped', 'librccl' in open('/proc/self/maps').read(), flush=True)\n"
             "    return out\n"
             "b.gather_counters = g\n"
             "from oadp_amd.oake.globals import Validator\n"
             f"Validator.main(['n1', {str(cfg)!r}])\n")
    r = subprocess.run([sys.executable, '-c', probe], env=env, cwd=str(root), capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count('GATHER backend nccl world 1 ranks 1 librccl mapped True') == 2, out[-2000:]  # val, then train
    assert '[n1] train: 24 images, 24 crops' in out and 'over 1 rank(s)' in out, out[-2000:]
    assert len(list((root / 'oake' / 'globals' / 'train2017').glob('*.pth'))) == 24
