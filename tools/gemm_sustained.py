"""One GEMM shape launched back to back for seconds (the board at its power cap, as in the bench) — GPU box only.
gemm_ablate.py times ten launches from idle clocks: that measures cycles.  This one measures what the bench sees.
usage: gemm_sustained.py [variants=4,13] [seconds=2.0] [rows=25600] [n=3072] [k=768] [mode=1 (gelu)]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load_lab()
dev = torch.device('cuda:0')
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,13').split(',')]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
m = int(sys.argv[3]) if len(sys.argv) > 3 else 25600
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3072
k = int(sys.argv[5]) if len(sys.argv) > 5 else 768
code = int(sys.argv[6]) if len(sys.argv) > 6 else 1
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
a = (torch.randn(m, k, device=dev) * 0.5).half()
w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
bias = torch.randn(n, device=dev)
c = torch.empty(m, n, device=dev, dtype=torch.float16)
for rnd in range(2):
    for v in variants:
        lib.oake_debug_set_gemm_variant(v)
        def burst(cnt):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(cnt):
                rc = lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, code, s)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1000 / cnt
        torch.cuda.synchronize(); time.sleep(0.5)
        first = burst(10)
        t0 = time.time(); series = []
        while time.time() - t0 < secs:
            series.append(burst(500))
        print(f'M{m} N{n} K{k} mode {code} variant {v} round {rnd}: first ten {first:6.1f} us, sustained '
              f'{" ".join(f"{x:.1f}" for x in series[:3])} ... last {series[-1]:6.1f} us '
              f'({2 * m * n * k / series[-1] / 1e6:5.0f} TF)', flush=True)
lib.oake_debug_set_gemm_variant(-1)
