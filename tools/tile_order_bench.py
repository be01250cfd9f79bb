"""A/B the GEMM tile order (oake_debug_set_gemm_panel) on the 16-bit-output shapes, one process."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib

lib = _lib.load()
dev = torch.device('cuda:0')
SHAPES = [('qkv', 12800, 2304, 768, 0), ('c_fc', 12800, 3072, 768, 1)]
PANELS = [int(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0, 3, 4, 6, 12, -10, -5, -20]
reps = 40
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, m, n, k, gelu in SHAPES:
    a = (torch.randn(m, k, device=dev) * 0.5).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev, dtype=torch.float16)
    ref = None
    for rnd in range(2):
        for p in PANELS:
            lib.oake_debug_set_gemm_panel(p)
            def run():
                assert lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, _lib.OAKE_F16, gelu, s) == 0
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            if ref is None:
                ref = c.clone()
            same = torch.equal(ref, c)
            print(f'{name:5s} panel {p:4d}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TFLOP/s  bit-identical {same}', flush=True)
lib.oake_debug_set_gemm_panel(0)
