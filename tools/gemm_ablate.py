"""What a GEMM tile costs without (parts of) its epilogue — GPU box only.
Runs the persistent kernel on the LN-free 16-bit epilogues in four forms on the K = 768 shapes:
    bias   acc + bias -> 16 bit                       (qkv's arithmetic)
    gelu   QuickGELU(acc + bias) -> 16 bit            (c_fc's arithmetic)
    raw    acc -> 16 bit (pack + trickled stores only)
    none   no epilogue at all (accumulators kept live, nothing stored)
interleaved in one process (rounds x forms), median per form.
usage: gemm_ablate.py [variants=4,8] [rounds=5] [rows=12800]"""
import ctypes as C, sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load_lab()  # the build that carries every variant
dev = torch.device('cuda:0')
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,8').split(',')]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
MODES = (('bias', 0), ('gelu', 1), ('raw', 3), ('none', 2))
m = int(sys.argv[3]) if len(sys.argv) > 3 else 12800
for name, n, k in (('c_fc', 3072, 768), ('qkv', 2304, 768), ('out_proj', 768, 768), ('c_proj', 768, 3072)):
    a = (torch.randn(m, k, device=dev) * 0.5).half()
    w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev, dtype=torch.float16)
    res = {}
    for rnd in range(rounds):
        for v in variants:
            lib.oake_debug_set_gemm_variant(v)
            for mode, code in MODES:
                def run():
                    rc = lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, code, s)
                    assert rc == 0, rc
                for _ in range(2): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10): run()
                e1.record(); torch.cuda.synchronize()
                res.setdefault((v, mode), []).append(e0.elapsed_time(e1) * 100)
    for v in variants:
        line = f'{name:7s} M{m} N{n} K{k} variant {v}:'
        for mode, _ in MODES:
            us = statistics.median(res[(v, mode)])
            line += f'  {mode} {us:6.1f} us ({2*m*n*k/us/1e6:5.0f} TF)'
        print(line, flush=True)
lib.oake_debug_set_gemm_variant(-1)
