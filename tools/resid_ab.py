"""A/B of GEMM variants on the two residual shapes (out_proj, c_proj; 16-bit residual epilogue), rounds
interleaved in one process.  usage: resid_ab.py [variants=4,9] [rounds=7]"""
import ctypes as C, sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load_lab()  # the build that carries every variant
dev = torch.device('cuda:0')
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,9').split(',')]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
m = 12800
for name, n, k in (('out_proj', 768, 768), ('c_proj', 768, 3072)):
    a = (torch.randn(m + 1, k, device=dev) * 0.5).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    bias = torch.randn(n, device=dev)
    x0 = torch.randn(m + 1, n, device=dev).half()
    part = torch.zeros((m + 1) * 32, device=dev)
    res, outs = {}, {}
    for rnd in range(rounds):
        for v in variants:
            lib.oake_debug_set_gemm_variant(v)
            x = x0.clone()
            def run():
                assert lib.oake_debug_gemm_resid16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(),
                                                   part.data_ptr(), m, n, k, 1, s) == 0
            run()
            torch.cuda.synchronize()
            outs[v] = (x.clone(), part.clone())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) * 50)
    same = all(torch.equal(outs[v][0], outs[variants[0]][0]) and torch.equal(outs[v][1], outs[variants[0]][1]) for v in variants)
    print(f'{name:9s} M{m} N{n} K{k}: ' + '  '.join(f'v{v} {statistics.median(res[v]):6.1f} us ({2*m*n*k/statistics.median(res[v])/1e6:5.0f} TF)' for v in variants)
          + f'  bit-identical {same}', flush=True)
lib.oake_debug_set_gemm_variant(-1)
