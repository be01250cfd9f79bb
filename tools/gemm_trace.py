"""Per-K-tile cycle anatomy of the GEMM main loop (debug trace, GPU box only)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m, n, k = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (12800, 768, 3072)
lib.oake_debug_set_gemm_variant(variant)
a = (torch.randn(m, k, device=dev) * 0.5).half(); w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
bias = torch.randn(n, device=dev); c = torch.empty(m, n, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, s)
trace = torch.zeros(16 * 64 * 4, dtype=torch.int64, device=dev)
lib.oake_debug_set_gemm_trace(C.c_void_p(trace.data_ptr()))
lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, s)
torch.cuda.synchronize()
lib.oake_debug_set_gemm_trace(None)
t = trace.view(16, 64, 4).cpu()
nk = min(k // 64 - 1, 64)
for b in (0, 1, 8):
    tb = t[b, :nk]
    issue = (tb[:, 1] - tb[:, 0]).float(); comp = (tb[:, 2] - tb[:, 1]).float(); bar = (tb[:, 3] - tb[:, 2]).float()
    tot = (tb[1:, 0] - tb[:-1, 0]).float()
    print(f'variant {variant} M{m} N{n} K{k} block {b}: per K-tile cycles: issue {issue.mean():.0f}  compute {comp.mean():.0f}  barrier-wait {bar.mean():.0f}  total {tot.mean():.0f}')
    print('   first 12 iters total:', [int(x) for x in tot[:12]], ' barrier:', [int(x) for x in bar[:12]])
