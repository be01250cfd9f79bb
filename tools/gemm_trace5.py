"""Phase anatomy of the ping-pong GEMM (variant 6): LOAD0 / MFMA0 / LOAD1 / MFMA1 cycles."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m, n, k = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (12800, 768, 3072)
lib.oake_debug_set_gemm_variant(variant)
a = (torch.randn(m, k, device=dev) * 0.5).half(); w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
bias = torch.randn(n, device=dev); c = torch.empty(m, n, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, s)
trace = torch.zeros(16 * 2 * 32 * 4, dtype=torch.int64, device=dev)
lib.oake_debug_set_gemm_trace(C.c_void_p(trace.data_ptr()))
lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, s)
torch.cuda.synchronize()
lib.oake_debug_set_gemm_trace(None)
t = trace.view(16, 2, 32, 4).cpu()
nk = min(k // 64, 32)
for b in (0, 9):
    for grp in (0, 1):
        tb = t[b, grp, :nk]
        l0 = (tb[:, 1] - tb[:, 0]).float(); m0 = (tb[:, 2] - tb[:, 1]).float(); l1 = (tb[:, 3] - tb[:, 2]).float()
        m1 = (tb[1:, 0] - tb[:-1, 3]).float(); tot = (tb[1:, 0] - tb[:-1, 0]).float()
        print(f'v{variant} M{m} N{n} K{k} block {b} group {"early" if grp == 0 else "late"}: LOAD0 {l0[1:].mean():.0f}  MFMA0 {m0[1:].mean():.0f}  LOAD1 {l1[1:].mean():.0f}  MFMA1 {m1.mean():.0f}  per K-tile {tot.mean():.0f}')
        print('    totals:', [int(x) for x in tot[:11]])
