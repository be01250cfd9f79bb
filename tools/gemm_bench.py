"""Time the GEMM kernel on the encoder's shapes through the C ABI (GPU box only).
usage: gemm_bench.py [variant] [shape-name-filter] [reps]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib

lib = _lib.load()
dev = torch.device('cuda:0')
SHAPES = [('qkv', 12800, 2304, 768), ('out_proj', 12800, 768, 768), ('c_fc', 12800, 3072, 768),
          ('c_proj', 12800, 768, 3072), ('conv1', 12544, 768, 3072), ('sq4096', 4096, 4096, 4096)]
variant = int(sys.argv[1]) if len(sys.argv) > 1 else -1
filt = sys.argv[2] if len(sys.argv) > 2 else ''
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lib.oake_debug_set_gemm_variant(variant)
for name, m, n, k in SHAPES:
    if filt and filt not in name:
        continue
    a = (torch.randn(m, k, device=dev) * 0.5).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        assert lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, _lib.OAKE_F16, s) == 0
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
    ref = a.float() @ w.float().t() + bias
    err = (c - ref).abs().max().item()
    print(f'variant {variant} {name:9s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TFLOP/s  maxerr {err:.2e}', flush=True)
