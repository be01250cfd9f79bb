"""Environment flags, as ``todd.Store`` exposes them to the reference:
DRY_RUN (quick integrity check), CUDA / CPU (device presence)."""
from __future__ import annotations

import os
import sys

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
    """Contiguous share of the host's logical CPUs for one of `local_world` ranks on this node (the fallback when
    sysfs says nothing about the topology)."""
    n = len(cpus)
    if local_world <= 1 or n < local_world:
        return list(cpus)
    lo, hi = local_rank * n // local_world, (local_rank + 1) * n // local_world
    return list(cpus[lo:hi])


def _parse_cpulist(text: str) -> list[int]:
    """sysfs cpulist syntax: "0-63,128-191"."""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _fmt_cpulist(cpus: list[int]) -> str:
    runs, cpus = [], sorted(cpus)
    for c in cpus:
        if runs and c == runs[-1][1] + 1:
            runs[-1][1] = c
        else:
            runs.append([c, c])
    return ','.join(f'{a}-{b}' if b > a else f'{a}' for a, b in runs)


def _read(path: str) -> str | None:
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def local_ranks(env=None) -> tuple[int, int] | None:
    """(local rank, ranks on THIS host) — only where that is actually known: LOCAL_RANK + LOCAL_WORLD_SIZE (torchrun
    exports both), or OAKE_SHARD=r/W together with an explicit OAKE_LOCAL_SHARDS=n (n shards of the W run on this
    host; local rank = r mod n).  A lone `OAKE_SHARD=0/8` process, a one-process-per-node array job, or a launcher
    that exports only the global WORLD_SIZE returns None: slicing the host by a count that is not this host's would
    make the rank host-bound on 1/W of the cores (advisor r04)."""
    env = os.environ if env is None else env
    if 'LOCAL_RANK' in env and env.get('LOCAL_WORLD_SIZE'):
        return int(env['LOCAL_RANK']), int(env['LOCAL_WORLD_SIZE'])
    shard = parse_shard(env)
    if shard and env.get('OAKE_LOCAL_SHARDS'):
        n = int(env['OAKE_LOCAL_SHARDS'])
        if n >= 1:
            return shard[0] % n, n
    return None


def gpu_pci_addresses() -> list[str] | None:
    """PCI addresses ("dddd:bb:dd.f") of the visible GPUs in HIP device order, or None without a HIP device.
    (/sys/class/drm/card<i> is NOT in HIP order, and a container sees every card's node: the PCI address is the key.)"""
    if not torch.cuda.is_available():
        return None
    out = []
    for i in range(torch.cuda.device_count()):
        p = torch.cuda.get_device_properties(i)
        try:
            out.append(f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0')
        except AttributeError:
            return None
    return out


def plan_cpus(local_rank: int, local_world: int, allowed: list[int], gpu_pci: list[str] | None = None,
              sysfs: str = '/sys', device_index: int | None = None) -> tuple[list[int], str]:
    """The CPUs rank `local_rank` of a node's `local_world` ranks should run on, and how they were chosen.

    Topology-aware: rank r drives GPU r mod #GPUs; that GPU's NUMA node is read from
    <sysfs>/bus/pci/devices/<addr>/numa_node and the node's CPUs from <sysfs>/devices/system/node/node<n>/cpulist;
    the node's CPUs (those this process may run on) are split among the ranks whose GPUs share the node, whole
    cores at a time — SMT siblings (<sysfs>/devices/system/cpu/cpu<c>/topology/thread_siblings_list) stay together,
    whatever the numbering (siblings at +N/2 is the usual one).  On a two-socket host ranks 4-7 then sit on the
    socket that owns GPUs 4-7, which a split of the logical CPU ids by rank does not give.  Where sysfs is silent
    (no numa_node, node -1, no cpulist): the contiguous slices of `cpu_slice`.

    `device_index` (advisor r05): the HIP device this rank actually drives, where that is not local_rank mod #GPUs —
    OAKE_LOCAL_SHARDS=n with n below the GPU count picks the device by the GLOBAL shard (shard_device_index).  With the
    visibility narrowed to ONE GPU per rank (HIP_VISIBLE_DEVICES set per rank: every rank sees its own GPU as device 0)
    the other ranks' GPUs are unknown: the rank then takes a share of its GPU's node sized for local_world / #nodes
    ranks (ranks of a node assumed contiguous in rank order) instead of 1 / local_world of it, which left half of each
    socket idle on a two-socket host."""
    allowed = sorted(allowed)
    if local_world <= 1:
        return allowed, 'single rank: affinity unchanged'
    fallback = (cpu_slice(local_rank % local_world, local_world, allowed), 'contiguous slice of the allowed CPUs')
    if not gpu_pci:
        return fallback
    ng = len(gpu_pci)

    def node_of(addr: str) -> int | None:
        t = _read(f'{sysfs}/bus/pci/devices/{addr}/numa_node')
        try:
            n = int(t) if t is not None else -1
        except ValueError:
            n = -1
        return n if n >= 0 else None

    nodes = [node_of(a) for a in gpu_pci]
    dev = (device_index if device_index is not None else local_rank) % ng
    mine = nodes[dev]
    if mine is None:
        return fallback
    t = _read(f'{sysfs}/devices/system/node/node{mine}/cpulist')
    if not t:
        return fallback
    node_cpus = [c for c in _parse_cpulist(t) if c in set(allowed)]
    if ng == 1 and local_world > 1:
        # one visible GPU per rank: who shares its node cannot be read off the device list
        n_nodes = 0
        while _read(f'{sysfs}/devices/system/node/node{n_nodes}/cpulist'):
            n_nodes += 1
        per_node = -(-local_world // max(n_nodes, 1))
        sharing = list(range(per_node))
        k_of = local_rank % per_node
    else:
        # the device each local rank drives: its own where the caller named it, r mod #GPUs for the others
        sharing = [r for r in range(local_world) if nodes[(dev if r == local_rank else r) % ng] == mine]
        k_of = sharing.index(local_rank)
    if not node_cpus or len(node_cpus) < len(sharing):
        return fallback
    # whole cores: a core = the set of its hardware threads that are in node_cpus
    seen, cores = set(), []
    for c in node_cpus:
        if c in seen:
            continue
        sib = _read(f'{sysfs}/devices/system/cpu/cpu{c}/topology/thread_siblings_list')
        grp = [s for s in (_parse_cpulist(sib) if sib else [c]) if s in set(node_cpus) and s not in seen] or [c]
        seen.update(grp)
        cores.append(sorted(grp))
    k, m = k_of, len(sharing)
    if len(cores) < m:
        return fallback
    part = cores[k * len(cores) // m:(k + 1) * len(cores) // m]
    keep = sorted(c for core in part for c in core)
    return keep, f'NUMA node {mine} of GPU {dev} ({gpu_pci[dev]}), share {k + 1} of {m}, whole cores'


def pin_cpus(env=None, gpu_pci: list[str] | None = None, sysfs: str = '/sys', apply: bool = True) -> list[int] | None:
    """Per-rank CPU affinity for a multi-rank node: a rank's host side (file reads, Huffman threads, index math, .pth
    writers: ~11 cores per rank at full rate, docs/history/round3.md item 9) stays on cores of the socket its GPU hangs off
    (`plan_cpus`) instead of migrating across a 128-256-thread host.  Done ONLY where the number of ranks on this host
    is known (`local_ranks`); OAKE_CPU_AFFINITY=0 switches it off.  The CPUs kept are printed once per process.
    Returns the CPUs kept (None: nothing done)."""
    env = os.environ if env is None else env
    if env.get('OAKE_CPU_AFFINITY', '1').lower() in ('0', 'false', 'no', 'off') or not hasattr(os, 'sched_setaffinity'):
        return None
    lr_lw = local_ranks(env)
    if lr_lw is None or lr_lw[1] <= 1:
        return None
    lr, lw = lr_lw
    cpus = sorted(os.sched_getaffinity(0))
    if gpu_pci is None:
        try:
            gpu_pci = gpu_pci_addresses()
        except Exception:  # noqa: BLE001 — topology is an optimisation, never a reason to fail a run
            gpu_pci = None
    # the device this rank really drives (OAKE_LOCAL_SHARDS below the GPU count: by the GLOBAL shard, shard_device_index)
    dev = shard_device_index(len(gpu_pci), env) if gpu_pci else None
    keep, how = plan_cpus(lr % lw, lw, cpus, gpu_pci, sysfs, device_index=dev)
    if not keep or len(keep) == len(cpus):
        return None
    if apply:
        os.sched_setaffinity(0, keep)
        print(f'[oake] local rank {lr}/{lw}: CPUs {_fmt_cpulist(keep)} ({len(keep)} of {len(cpus)}; {how}; '
              f'OAKE_CPU_AFFINITY=0 to disable)', file=sys.stderr, flush=True)
    return keep
