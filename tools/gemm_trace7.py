"""Whole-kernel anatomy of the loader-wave GEMM (variant 10): entry -> loop -> epilogue -> exit."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
m, n, k = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (12800, 768, 3072)
lib.oake_debug_set_gemm_variant(10)
a = (torch.randn(m, k, device=dev) * 0.5).half(); w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
bias = torch.randn(n, device=dev); c = torch.empty(m, n, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, s)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); run(); e1.record(); torch.cuda.synchronize()
single = e0.elapsed_time(e1) * 1e3
trace = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
lib.oake_debug_set_gemm_trace(C.c_void_p(trace.data_ptr()))
e0.record(); run(); e1.record(); torch.cuda.synchronize()
lib.oake_debug_set_gemm_trace(None)
t = trace.view(256, 4).cpu()
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
print(f'M{m} N{n} K{k}: single launch {single:.1f} us (traced {e0.elapsed_time(e1)*1e3:.1f} us); blocks traced {len(t)}')
print(f'  cycles: entry spread {int((t[:,0]-t0).max())}  prologue {float((t[:,1]-t[:,0]).float().mean()):.0f}  loop {float((t[:,2]-t[:,1]).float().mean()):.0f} ({float((t[:,2]-t[:,1]).float().mean())/(k//64):.0f}/K-tile)  epilogue {float((t[:,3]-t[:,2]).float().mean()):.0f}  first-entry->last-exit {int(t[:,3].max()-t0)}')
