"""Where does a workgroup of attn_out_kernel (csrc/attn_out.hip) spend its cycles?  Runs the s_memtime-stamped
measurement build on a batch of 256 images x 50 tokens and prints, per phase boundary, the cycles since the previous
one (median over the 8 waves of the first 4 workgroups), plus the launch time of the production build.
usage (GPU box): python tools/attn_out_trace.py [n=256] [L=50]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, '.')
from oadp_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = _lib.load_lab()  # the build that carries every variant
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
c = 768
qkv = torch.randn(n * L, 3 * c, generator=g)
qkv[:, :c] *= 0.35
qkv = qkv.half().to(dev)
w = (torch.randn(c, c, generator=g) * c ** -0.5).half().to(dev)
bias = torch.randn(c, generator=g).to(dev)
x = torch.zeros(n * L, c, dtype=torch.float16, device=dev)
part = torch.zeros(n * L, 16, 2, device=dev)
trace = torch.zeros(4 * 12 * 64, dtype=torch.int64, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(tr, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = lib.oake_debug_attn_out_trace(qkv.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), part.data_ptr(),
                                       n, L, 12, _lib.OAKE_F16, tr, reps, st)
    assert rc == 0, rc
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


run(None, 50)
print(f'production build: {run(None, 400):.2f} us per launch (n = {n}, L = {L}; includes ~1 us of launch gap)')
run(trace.data_ptr(), 20)
print(f'stamped build:    {run(trace.data_ptr(), 200):.2f} us per launch')
t = trace.view(4, 12, 64).cpu()
ATT, OUT = [8, 9, 10, 11], list(range(8))  # wave ids by role (csrc/attn_out.hip)


def report(title, waves, names):
    print(title)
    pts = [p for p in sorted(names) if (t[:, waves, p] != 0).all()]
    prev = None
    for p in pts:
        if prev is not None:
            d = (t[:, waves, p] - t[:, waves, prev]).flatten().float()
            print(f'{p:3d} {names[p]:48s} median {d.median().item():8.0f}  min {d.min().item():8.0f}  max {d.max().item():8.0f}')
        prev = p
    span = (t[:, waves, pts[-1]] - t[:, waves, pts[0]]).flatten().float()
    print(f'    entry -> last stamp: median {span.median().item():.0f} cycles (s_memtime counts shader-clock cycles here)')


na = {0: 'entry', 1: 'rows of unit 0 requested and landed'}
for u in range(6):
    na[2 + 8 * u] = f'unit {u}: barrier'
    na[3 + 8 * u] = f'unit {u}: V reads, next rows requested, scores'
    na[4 + 8 * u] = f'unit {u}: softmax (2 query tiles)'
    na[5 + 8 * u] = f'unit {u}: P V + O fragments written'
    na[6 + 8 * u] = f'unit {u}: wait (O written, next rows landed)'
report('attention waves', ATT, na)
no = {0: 'entry', 1: 'first W fragments requested', 2: 'barriers (rows of unit 0; O of unit 0)'}
for s_ in range(6):
    no[3 + 2 * s_] = f'step {s_}: 4 groups x 24 MFMAs'
    no[4 + 2 * s_] = f'step {s_}: barrier'
no[40] = 'epilogue (residual, stores, stats)'
no[41] = 'stats of the shared slices'
report('out_proj waves', OUT, no)
