"""Environment flags, as ``todd.Store`` exposes them to the reference:
DRY_RUN (quick integrity check), CUDA / CPU (device presence)."""
from __future__ import annotations

import os

import torch


def _flag(name: str) -> bool:
    return os.environ.get(name, '').lower() in ('1', 'true', 'yes', 'on')


class _Store:

    @property
    def DRY_RUN(self) -> bool:
        return _flag('DRY_RUN')

    @property
    def CUDA(self) -> bool:
        return torch.cuda.is_available()

    @property
    def CPU(self) -> bool:
        return not self.CUDA


Store = _Store()


def get_rank() -> int:
    return int(os.environ.get('RANK', 0))


def get_local_rank() -> int:
    return int(os.environ.get('LOCAL_RANK', 0))


def get_world_size() -> int:
    return int(os.environ.get('WORLD_SIZE', 1))


def parse_shard(env=None) -> tuple[int, int] | None:
    """``OAKE_SHARD=r/W``: this process is DistributedSampler shard r of W without a process group (array-job style
    launches; the path has no data-path collective).  Returns (r, W) or None; malformed values raise a ValueError
    that says what the format is."""
    env = os.environ if env is None else env
    v = env.get('OAKE_SHARD')
    if not v:
        return None
    try:
        r, w = (int(t) for t in v.split('/'))
    except ValueError:
        raise ValueError(f'OAKE_SHARD={v!r}: expected "r/W" with integers 0 <= r < W, e.g. OAKE_SHARD=3/8') from None
    if not 0 <= r < w:
        raise ValueError(f'OAKE_SHARD={v!r}: need 0 <= r < W')
    return r, w


def shard_device_index(gpus: int, env=None) -> int:
    """The GPU this process drives: LOCAL_RANK mod the visible GPUs under a launcher; under OAKE_SHARD=r/W without
    a launcher r mod the visible GPUs (eight shards started on one node land on eight GPUs without per-process
    HIP_VISIBLE_DEVICES; with HIP_VISIBLE_DEVICES narrowed to one device it is device 0 either way)."""
    env = os.environ if env is None else env
    if 'LOCAL_RANK' in env:
        return int(env['LOCAL_RANK']) % max(gpus, 1)
    shard = parse_shard(env)
    return (shard[0] if shard else 0) % max(gpus, 1)


def cpu_slice(local_rank: int, local_world: int, cpus: list[int]) -> list[int]:
    """Contiguous share of the host's logical CPUs for one of `local_world` ranks on this node."""
    n = len(cpus)
    if local_world <= 1 or n < local_world:
        return list(cpus)
    lo, hi = local_rank * n // local_world, (local_rank + 1) * n // local_world
    return list(cpus[lo:hi])


def pin_cpus(env=None) -> list[int] | None:
    """Per-rank CPU affinity for a multi-rank node: rank i of the node's N ranks (LOCAL_RANK / LOCAL_WORLD_SIZE, or
    the OAKE_SHARD pair) keeps the i-th contiguous N-th of the CPUs this process may run on — a rank's host side
    (file reads, Huffman threads, index math, .pth writers: ~11 cores per rank at full rate)
    then stays on its own cores and caches instead of migrating across a 128-256-thread host.  OAKE_CPU_AFFINITY=0
    switches it off.  Returns the CPUs kept (None: nothing done)."""
    env = os.environ if env is None else env
    if env.get('OAKE_CPU_AFFINITY', '1').lower() in ('0', 'false', 'no', 'off') or not hasattr(os, 'sched_setaffinity'):
        return None
    shard = parse_shard(env)
    if 'LOCAL_RANK' in env:
        lr, lw = int(env['LOCAL_RANK']), int(env.get('LOCAL_WORLD_SIZE') or env.get('WORLD_SIZE') or 1)
    elif shard and shard[1] <= 16:  # (a larger W spans nodes: how many shards share this host is not knowable here)
        lr, lw = shard
    else:
        return None
    if lw <= 1:
        return None
    cpus = sorted(os.sched_getaffinity(0))
    keep = cpu_slice(lr % lw, lw, cpus)
    if not keep or len(keep) == len(cpus):
        return None
    os.sched_setaffinity(0, keep)
    return keep
