"""Vendor yardstick (tools only — never in the product path): what do the vendor GEMMs under torch
(rocBLAS / hipBLASLt) reach on the encoder's four GEMM shapes, on the same board, in the same session, with the
same random f16 data and under the same power cap as our kernels?  Answers whether ~1150 TFLOP/s for a bare
K = 768 loop is near what this board gives at that K or whether the loop has a large margin left
("next round" item 2).

Per shape, interleaved in one process (rounds x forms), median:
    torch    torch.matmul(A, W.T) -> f16                 (no bias, no epilogue: compare with ours 'none' / 'raw')
    linear   torch.nn.functional.linear(A, W, bias)      (bias epilogue)
    ours     oake_debug_gemm16: bias / raw / none forms of the persistent kernel (as tools/gemm_ablate.py)
usage: vendor_gemm_yardstick.py [rounds=5]      (TORCH_BLAS_PREFER_HIPBLASLT=0|1 selects the torch backend)"""
import ctypes as C
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib

lib = _lib.load_lab()  # the build that carries every variant (measurement epilogues)
dev = torch.device('cuda:0')
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
m = 12800
try:
    backend = torch.backends.cuda.preferred_blas_library()
except Exception:  # noqa: BLE001
    backend = 'unknown'
print(json.dumps({'torch': torch.__version__, 'blas_backend': str(backend),
                  'TORCH_BLAS_PREFER_HIPBLASLT': os.environ.get('TORCH_BLAS_PREFER_HIPBLASLT')}), flush=True)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us per call


for name, n, k in (('c_fc', 3072, 768), ('qkv', 2304, 768), ('out_proj', 768, 768), ('c_proj', 768, 3072)):
    a = (torch.randn(m, k, device=dev) * 0.5).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    bias = torch.randn(n, device=dev)
    bias16 = bias.half()
    c = torch.empty(m, n, device=dev, dtype=torch.float16)
    wt = w.t()
    forms = {
        'torch_matmul': lambda: torch.matmul(a, wt, out=c),
        'torch_linear_bias': lambda: torch.nn.functional.linear(a, w, bias16),
    }
    for label, code in (('ours_bias', 0), ('ours_raw', 3), ('ours_none', 2)):
        def run(code=code):
            rc = lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, code, s)
            assert rc == 0, rc
        forms[label] = run
    res = {f: [] for f in forms}
    for _ in range(rounds):
        for f, fn in forms.items():
            res[f].append(timed(fn))
    # sanity: the vendor result equals ours to fp16 rounding
    forms['ours_bias']()
    ref = torch.nn.functional.linear(a, w, bias16)
    err = (c.float() - ref.float()).abs().max().item()
    out = {'shape': name, 'M': m, 'N': n, 'K': k, 'max_abs_diff_ours_vs_vendor': round(err, 5)}
    for f in forms:
        us = statistics.median(res[f])
        out[f] = {'us': round(us, 2), 'tflops': round(2 * m * n * k / us / 1e6, 1)}
    print(json.dumps(out), flush=True)
