"""Stand-alone timings of the four production GEMM shapes of a ViT-B/32 layer at 256 images
(M = 12800): qkv (LN-folded), out-proj (residual), c_fc (LN-folded + GELU), c_proj (residual).
GPU box only.  usage: gemm_shapes.py [M]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
m = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


total = 0.0
for name, n, k, kind in (('qkv', 2304, 768, 'ln'), ('out_proj', 768, 768, 'resid'),
                         ('c_fc', 3072, 768, 'gelu'), ('c_proj', 768, 3072, 'resid')):
    a = (torch.randn(m + 1, k, device=dev) * 0.5).half()
    w32 = torch.randn(n, k, device=dev) * k ** -0.5
    w = w32.half()
    bias = torch.randn(n, device=dev)
    if kind == 'resid':
        x = torch.randn(m + 1, n, device=dev).half()
        part = torch.zeros((m + 1) * 32, device=dev)
        fn = lambda: lib.oake_debug_gemm_resid16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(),
                                                 part.data_ptr(), m, n, k, 1, s)
    else:
        # the plain bias / GELU epilogues: the LN-folded debug entry folds weights and computes row
        # statistics on every call, which is not what the encoder's steady state does
        c = torch.empty(m, n, device=dev, dtype=torch.float16)
        fn = lambda: lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(),
                                           m, n, k, 1, int(kind == 'gelu'), s)
    us = timeit(fn)
    total += us
    print(f'{name:9s} M{m} N{n} K{k}: {us:7.1f} us = {2*m*n*k/us/1e6:6.0f} TFLOP/s')
print(f'sum {total:.1f} us per layer (GEMMs only)')
