"""bench.py's launcher contract: `python bench.py --gpus N` really runs N ranks
(reference: torchrun --nproc_per_node=${GPUS}, README.md:197-207; oadp/oake/base.py:122-126)."""
import json
import os
import pathlib
import subprocess
import sys

import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _run(args, env_extra, timeout=600):
    env = dict(os.environ, PYTHONPATH=str(ROOT), **env_extra)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'OAKE_BENCH_FULL_LINE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *args], capture_output=True, text=True,
                       env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks_cpu_plumbing():
    """No GPU here: the plumbing mode skips every GPU call but runs the real launcher, rendezvous,
    barrier, max-over-ranks reduction and counters gather over gloo."""
    line = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-profile'],
                dict(OAKE_BENCH_DRY_PLUMBING='1'))
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['warmup'] == 1
    assert 'ranks gathered: 2' in line['config']['sharding']
    assert line['value'] is None and 'dry-run' in line['data']  # carries no throughput claim
    assert line['cpu_baseline'] is None


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, PYTHONPATH=str(ROOT), OAKE_BENCH_DRY_PLUMBING='1', WORLD_SIZE='1', RANK='0')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '1', '--no-profile'],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_gpus_2_on_one_gpu_over_gloo():
    """Two ranks sharing the one GPU of the test box (gloo instead of RCCL): the N > 1 path end to end."""
    line = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--no-profile', '--no-cpu-baseline', '--batch', '32'],
                dict(OAKE_BENCH_BACKEND='gloo'))
    assert line['n_gpus'] == 2 and line['value'] > 0
    assert 'ranks gathered: 2' in line['config']['sharding']
    assert abs(line['crops_per_sec'] - line['value']) < 1e-6 * line['value'] + 1  # globals: one crop per image


@pytest.mark.gpu
@pytest.mark.parametrize('mode,extra', [('blocks', ['--batch', '4']), ('objects', ['--batch', '2', '--proposals', '40'])])
def test_modes_produce_a_contract_line(mode, extra):
    line = _run(['--mode', mode, '--steps', '2', '--warmup', '1', '--no-cpu-baseline', *extra], {})
    assert line['config']['mode'] == mode and line['n_gpus'] == 1 and line['value'] > 0
    assert line['unit'] == 'images/sec' and line['crops_per_sec'] > line['value']
    rf = line['roofline']
    assert rf['bound'] == 'mfma' and 0 < rf['frac'] < 1 and rf['kernel'].startswith('gemm')
    # the live power-cap probe: zero operands reach the data-sheet rate, random ones are capped below it
    sus = rf['board_mfma_tflops']
    assert sus['live'] and 1000 < sus['random_operands'] <= sus['zero_operands'] * 1.02 < 2700
    assert 0 < line['mfma_sustained_frac_e2e'] < 1 and 0 < rf['frac_of_board_random'] < 1
    # the calibration a reader needs sits in `roofline` itself; the per-kernel tables are in the side file
    assert rf['sum_le_stamped_wall'] and 0.8 < rf['sum_over_step'] < 1.1  # (tiny config: ratio only loosely bound)
    assert len(json.dumps(line)) < 7000 and line['detail'] and (ROOT / line['detail']).exists()
    full = json.loads((ROOT / line['detail']).read_text())
    assert full['value'] == line['value'] and full['kernels'] and full['roofline']['sustained']['live']


@pytest.mark.gpu
def test_two_validator_ranks_share_one_gpu(tmp_path):
    """More ranks than GPUs (docs/history/design_sections_5_6_as_of_round5.md §5.5: the host side of the sweep scales with processes): two full
    validators on the DistributedSampler halves of one image set, one GPU, gloo for the counters gather —
    every image gets its file exactly once."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    root = tmp_path / 'set'
    env = dict(os.environ, PYTHONPATH=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port),
                        str(ROOT / 'tools' / 'sweep_ranks.py'), '48', 'blocks', str(root)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if 'rank(s) on one GPU' in ln]
    assert len(line) == 1 and ' 2 rank(s)' in line[0] and ': 48 images, 1296 crops' in line[0], r.stdout[-2000:]
    files = sorted(p.name for p in (root / 'out_blocks_2').glob('*.pth'))
    assert files == [f'{i:012d}.pth' for i in range(48)]


def test_gpus_8_plumbing_eight_gloo_ranks():
    """BASELINE.json configs[3] / [4] launch shape without an 8-GPU node: `--gpus 8` re-executes under
    torch.distributed.run with eight ranks; rendezvous, the all_reduce that counts the ranks, barrier,
    max-over-ranks time and the 8 x 32-byte counters gather all run (gloo, no GPU work), ONE JSON line comes out."""
    line = _run(['--gpus', '8', '--steps', '2', '--warmup', '1', '--no-profile'],
                dict(OAKE_BENCH_DRY_PLUMBING='1'), timeout=900)
    assert line['n_gpus'] == 8 and line['steps'] == 2 and line['scaling'] == 'weak'
    assert 'images x8' in line['config']['sharding'] and 'ranks gathered: 8' in line['config']['sharding']
    assert line['value'] is None and line['cpu_baseline'] is None
