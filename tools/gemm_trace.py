"""Per-tile cycle anatomy of the production GEMM (gemm_pp_kernel) — GPU box only.
usage: gemm_trace.py M N K [gelu|bias|f32|resid]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load_lab()  # the build that carries every variant
dev = torch.device('cuda:0')
m, n, k = (int(v) for v in sys.argv[1:4])
mode = sys.argv[4] if len(sys.argv) > 4 else 'bias'
lib.oake_debug_set_gemm_variant(int(sys.argv[5]) if len(sys.argv) > 5 else 10)  # 10: four short phases (the stamped form)
a = (torch.randn(m, k, device=dev) * 0.5).half(); w = (torch.randn(n, k, device=dev) * k ** -0.5).half()
bias = torch.randn(n, device=dev)
c = torch.empty(m, n, device=dev, dtype=torch.float32 if mode == 'f32' else torch.float16)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
xres = torch.randn(m + 1, n, device=dev).half()
part = torch.zeros((m + 1) * 32, device=dev)
def run():
    if mode == 'resid':
        lib.oake_debug_gemm_resid16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), xres.data_ptr(), part.data_ptr(),
                                    m, n, k, 1, s)
    elif mode == 'f32':
        lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, s)
    else:
        lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, 1, int(mode == 'gelu'), s)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f'M{m} N{n} K{k} {mode}: {us:.1f} us/launch = {2*m*n*k/us/1e6:.0f} TFLOP/s', flush=True)
trace = torch.zeros(4096 + 512, dtype=torch.int64, device=dev)
lib.oake_debug_set_gemm_trace(C.c_void_p(trace.data_ptr()))
run(); torch.cuda.synchronize()
lib.oake_debug_set_gemm_trace(None)
if len(sys.argv) > 5 and int(sys.argv[5]) == 13:  # gemm_w8_kernel: per-phase sums (gemm_w8.inc W8_STAMP)
    t = trace[:2048].view(64, 2, 16).cpu()
    names = ('LOAD0', 'bar', 'MFMA0', 'bar', 'LOAD1', 'bar', 'MFMA1+wait', 'bar', 'entry+epilogues')
    for b in (0, 7, 40):
        for grp in (0, 1):
            r = t[b, grp]; nkt = int(r[15])
            if nkt == 0: continue
            per = [int(r[i]) / nkt for i in range(8)]
            print(f'  block {b} group {grp}: {nkt} K-tiles; cycles per K-tile ' + ' | '.join(f'{n} {c:.0f}' for n, c in zip(names, per))
                  + f' = {sum(per):.0f}; entry + epilogues {int(r[8])} in all')
    sys.exit(0)
wc = trace[4096:].view(256, 2).cpu(); wc = wc[wc[:, 0] > 0]
if len(wc) == 0:
    sys.exit(0)  # a kernel without cycle stamps
t = trace[:4096].view(64, 2, 8, 4).cpu()
span = (wc[:, 1].max() - wc[:, 0].min()).item() / 100.0
print(f'  wall-clock (100 MHz): first entry -> last exit {span:.1f} us; entry spread {(wc[:,0].max()-wc[:,0].min()).item()/100:.1f} us; exit spread {(wc[:,1].max()-wc[:,1].min()).item()/100:.1f} us; per-block durations min/mean/max {((wc[:,1]-wc[:,0]).min().item())/100:.1f}/{((wc[:,1]-wc[:,0]).float().mean().item())/100:.1f}/{((wc[:,1]-wc[:,0]).max().item())/100:.1f} us')
for b in (0, 7, 40):
    for grp in (0, 1):
        rows = t[b, grp]; rows = rows[rows[:, 3] > 0]
        if len(rows) == 0: continue
        loop = (rows[:, 2] - rows[:, 1]).tolist(); ep = (rows[:, 3] - rows[:, 2]).tolist()
        print(f'  block {b} grp {grp}: entry->first tile {int(rows[0,1]-rows[0,0])}; per tile loop {loop} ({[round(x/(k//64)) for x in loop]}/K-tile) epilogue {ep}; total {int(rows[-1,3]-rows[0,0])}')
