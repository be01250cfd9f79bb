"""Micro-benchmarks of the non-GEMM kernels through the C ABI debug entry points (GPU box only)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

def timeit(fn, reps=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

n, L, H = 256, 50, 12
qkv = torch.randn(n * L, 3 * H * 64, device=dev).half()
out = torch.empty(n * L, H * 64, device=dev, dtype=torch.float16)
for v in (0, 1, 2, 3, 31):
    lib.oake_debug_set_attention_variant(v)
    us = timeit(lambda: lib.oake_debug_attention(qkv.data_ptr(), out.data_ptr(), n, L, H, 1, s))
    print(f'attention variant {v}: {us:.1f} us  ({(qkv.numel()+out.numel())*2/us/1e6:.2f} TB/s)')
lib.oake_debug_set_attention_variant(31)
x = torch.randn(n * L, 768, device=dev); g = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
y = torch.empty(n * L, 768, device=dev, dtype=torch.float16)
us = timeit(lambda: lib.oake_debug_layernorm(x.data_ptr(), 0, g.data_ptr(), b.data_ptr(), y.data_ptr(), n * L, 768, 1, s))
print(f'layernorm: {us:.1f} us  ({(x.numel()*4+y.numel()*2)/us/1e6:.2f} TB/s)')
# cold variants: flush caches between launches with a big memset
junk = torch.empty(512 * 1024 * 1024, device=dev, dtype=torch.uint8)
def cold(fn):
    tot = 0.0
    for _ in range(5):
        junk.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); tot += e0.elapsed_time(e1)
    return tot / 5 * 1e3
print(f'attention cold: {cold(lambda: lib.oake_debug_attention(qkv.data_ptr(), out.data_ptr(), n, L, H, 1, s)):.1f} us')
print(f'layernorm cold: {cold(lambda: lib.oake_debug_layernorm(x.data_ptr(), 0, g.data_ptr(), b.data_ptr(), y.data_ptr(), n * L, 768, 1, s)):.1f} us')
