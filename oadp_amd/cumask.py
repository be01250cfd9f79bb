"""CU-masked HIP streams (``hipExtStreamCreateWithCUMask``) for running two lanes on disjoint halves of the chip.

Two lanes time-slicing the whole chip overlap each other's launch gaps and tails, but every kernel
still pays its own start-up (all CUs miss their first K-tiles at once) and its own tile-end burst (all CUs store
their last tile at once) with nothing running beside it.  On disjoint CU sets the two lanes' kernels are truly
concurrent: one lane's fixed per-kernel costs run beside the other lane's K loops.  Each lane takes half of EVERY
XCD, so block b of a launch still lands on XCD b % 8 (the kernels' XCD-aware tile order) and both lanes share
every XCD's fabric link instead of saturating four of them each.

The product path takes streams from its caller (``torch.cuda.current_stream``): this module only creates them.
"""
from __future__ import annotations

import ctypes as C

_hip = None


def hip_runtime() -> C.CDLL:
    """The libamdhip64 torch has already loaded (one runtime per process)."""
    global _hip
    if _hip is None:
        import torch  # noqa: F401
        path = None
        with open('/proc/self/maps') as f:
            for line in f:
                if 'libamdhip64' in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError('libamdhip64 is not loaded (import torch on a ROCm build first)')
        _hip = C.CDLL(path)
        _hip.hipExtStreamCreateWithCUMask.restype = C.c_int
        _hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    return _hip


def create_masked_stream(mask_words: list[int]) -> int:
    """A new HIP stream restricted to the CUs whose bits are set in ``mask_words`` (32 CUs per word); returns the
    raw ``hipStream_t`` (wrap with ``torch.cuda.ExternalStream``)."""
    hip = hip_runtime()
    arr = (C.c_uint32 * len(mask_words))(*mask_words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(mask_words), arr)
    if rc != 0:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask failed: {rc}')
    return s.value


def half_masks(n_cus: int, scheme: str = 'halves') -> list[list[int]]:
    """Two complementary masks over n_cus compute units.  scheme: 'halves' = bits [0, n/2) / [n/2, n) (the default:
    with mask bit i = XCD i % 8, CU i / 8 — tools/cu_mask_probe.py — that is half of EVERY XCD per lane);
    'even_odd' = even / odd bits; 'group<N>' = (i // N) even / odd ('pairs' is an alias of 'group8')."""
    if scheme == 'pairs':
        scheme = 'group8'
    def words(bits):
        w = [0] * ((n_cus + 31) // 32)
        for b in bits:
            w[b // 32] |= 1 << (b % 32)
        return w
    if scheme == 'halves':
        a = [i for i in range(n_cus) if i < n_cus // 2]
    elif scheme == 'even_odd':
        a = [i for i in range(n_cus) if i % 2 == 0]
    elif scheme.startswith('group'):
        g = int(scheme[5:])
        a = [i for i in range(n_cus) if (i // g) % 2 == 0]
    else:
        raise ValueError(scheme)
    b = [i for i in range(n_cus) if i not in set(a)]
    return [words(a), words(b)]
