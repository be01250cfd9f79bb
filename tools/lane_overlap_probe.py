"""Do two lanes really overlap a short, one-tile-per-CU GEMM with a long one?  
Lane 1 runs c_fc-shaped launches (M 12800, N 3072, K 768: 960 tiles, 3.75 per CU) back to back on its stream, lane 2
out_proj-shaped launches (N 768, K 768, residual epilogue: 240 tiles on 256 CUs) on another; both persistent kernels,
one block per CU.  Printed: each alone, both together (wall clock of n + n launches), and what perfect overlap
(max of the two) and none (their sum) would be.  GPU box only.   usage: lane_overlap_probe.py [n=200]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = 12800
a = (torch.randn(m, 768, device=dev) * 0.5).half()
w_fc = (torch.randn(3072, 768, device=dev) * 768 ** -0.5).half()
w_out = (torch.randn(768, 768, device=dev) * 768 ** -0.5).half()
b_fc, b_out = torch.randn(3072, device=dev), torch.randn(768, device=dev)
h = torch.empty(m, 3072, device=dev, dtype=torch.float16)
x = torch.randn(m + 1, 768, device=dev).half()
part = torch.zeros((m + 1) * 32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fc(s):
    assert lib.oake_debug_gemm16(a.data_ptr(), w_fc.data_ptr(), b_fc.data_ptr(), h.data_ptr(), m, 3072, 768, 1, 1,
                                 C.c_void_p(s.cuda_stream)) == 0


def out(s):
    assert lib.oake_debug_gemm_resid16(a.data_ptr(), w_out.data_ptr(), b_out.data_ptr(), x.data_ptr(), part.data_ptr(),
                                       m, 768, 768, 1, C.c_void_p(s.cuda_stream)) == 0


def timed(fn):
    for _ in range(3):
        fn(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rnd in range(3):
    t_fc = timed(lambda k: [fc(s1) for _ in range(k)])
    t_out = timed(lambda k: [out(s2) for _ in range(k)])
    t_both = timed(lambda k: [(fc(s1), out(s2)) for _ in range(k)])
    t_ser = timed(lambda k: [(fc(s1), out(s1)) for _ in range(k)])
    print(f'round {rnd}: c_fc alone {t_fc:.1f} us, out_proj alone {t_out:.1f} us per launch; one of each per iteration: '
          f'two lanes {t_both:.1f} us, one lane {t_ser:.1f} us  (perfect overlap {max(t_fc, t_out):.1f}, none {t_fc + t_out:.1f}; '
          f'recovered {(t_fc + t_out - t_both) / min(t_fc, t_out) * 100:.0f} % of the shorter kernel)', flush=True)
