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
