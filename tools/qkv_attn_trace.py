"""Where does a workgroup of qkv_attn_kernel (csrc/qkv_attn.hip) spend its cycles?  Cycle stamps of the first blocks'
tile phases (wave 0 = row group 0, wave 4 = row group 1, wave 8 = DMA wave 0) + the launch time.  GPU box only.
usage: python tools/qkv_attn_trace.py [n_img=256] [L=50] [quad]     (L >= 192: the objects-mode kernel, csrc/qkv_attn_obj.hip;
'quad': its four-images-per-tile form at L <= 50)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
l = int(sys.argv[2]) if len(sys.argv) > 2 else 50
heads, c = 12, 768
g = torch.Generator(device='cpu').manual_seed(1)
x = (torch.randn(n * l, c, generator=g) * 1.5).half().to(dev)
w = torch.randn(3 * c, c, generator=g) * c ** -0.5
w[:c] *= 0.125
w = w.to(dev)
gamma, beta, bias = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.zeros(3 * c, device=dev)
out = torch.empty(n * l, c, dtype=torch.float16, device=dev)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
OBJ = l >= 192
QUAD = len(sys.argv) > 3 and sys.argv[3] == 'quad'  # (l <= 50 through the 208-row tile kernel's four-image form)
if OBJ:
    x = (torch.randn(n * l + n, c, generator=g) * 1.5).half().to(dev)
    out = torch.empty(n * l + n, c, dtype=torch.float16, device=dev)
    mask = (torch.rand(n, l - 1, generator=g) < 0.4).half().to(dev)
def run(reps, trace=None):
    if OBJ:
        rc = lib.oake_debug_ln_qkv_attn_obj(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
                                            mask.data_ptr(), 1, out.data_ptr(), n, l, heads, 1, trace, reps, s)
        assert rc == 0, rc
        return
    rc = (lib.oake_debug_ln_qkv_attn_quad if QUAD else lib.oake_debug_ln_qkv_attn)(
        x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(), out.data_ptr(), n, l, heads, 1, trace, reps, s)
    assert rc == 0, rc
run(3)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(); run(20); e1.record(); torch.cuda.synchronize()
print(f'n {n} L {l}: {e0.elapsed_time(e1) * 50:.1f} us per launch incl. the debug entry\'s fold / permute passes / 20', flush=True)
if OBJ or QUAD:
    trace = torch.zeros(64 * 3 * 6 * 8, dtype=torch.int64, device=dev)
    run(1, C.c_void_p(trace.data_ptr()))
    t = trace.view(64, 3, 6, 8).cpu()
    names = ['K loop', 'q|k write', '-> X2', 'S + softmax', '-> X3', 'v write -> X4', 'PV + out -> X5']
    for b in (0, 1, 9, 40):
        for role, rn in ((0, 'group 0'), (1, 'group 1')):
            for i in [i for i in range(6) if t[b, role, i, 0] > 0]:
                r = t[b, role, i]
                d = [int(r[k + 1] - r[k]) for k in range(7)]
                print(f'  block {b:2d} {rn} tile {i}: ' + ', '.join(f'{nm} {v}' for nm, v in zip(names, d)) + f'; tile {int(r[7] - r[0])}')
        dn = ['S + softmax', '-> X3', 'X3 -> X4', 'PV + out']  # (the three waves of a SIMD share its issue slots:
        for i in [i for i in range(6) if t[b, 2, i, 0] > 0]:    #  the youngest wave's task takes the longest)
            r = t[b, 2, i]
            print(f'  block {b:2d} DMA wave 0 tile {i}: ' + ', '.join(f'{nm} {int(r[k + 1] - r[k])}' for k, nm in enumerate(dn)))
        print()
    sys.exit(0)
trace = torch.zeros(64 * 3 * 6 * 8, dtype=torch.int64, device=dev)
run(1, C.c_void_p(trace.data_ptr()))
t = trace.view(64, 3, 6, 8).cpu()
names = ['K loop', 'window write', '-> X2', 'attention', '-> X3']
for b in (0, 1, 8, 17, 40):
    for role, rn in ((0, 'group 0'), (1, 'group 1'), (2, 'DMA')):
        tiles = [i for i in range(6) if t[b, role, i, 0] > 0]
        for i in tiles:
            r = t[b, role, i]
            if role == 2:
                print(f'  block {b:2d} {rn:7s} tile {i}: staging span {int(r[1] - r[0]):6d}, X1..X3 {int(r[5] - r[1]):6d}')
            else:
                d = [int(r[k + 1] - r[k]) for k in range(5)]
                print(f'  block {b:2d} {rn:7s} tile {i}: ' + ', '.join(f'{nm} {v}' for nm, v in zip(names, d)) + f'; tile {int(r[5] - r[0])}')
    print()
