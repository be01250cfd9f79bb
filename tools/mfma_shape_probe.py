"""Which MFMA shape does the board sustain more of under its power cap?  Register-only streams (two waves per SIMD on every
CU, 160 accumulator registers per wave, no LDS, no memory) of v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16 on all-zero
and on N(0, 0.25) operands, each for `seconds`, interleaved — GPU box only.
usage: mfma_shape_probe.py [seconds=2.0] [rounds=2]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
lib = _lib.load()
dev = torch.device('cuda:0')
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
sink = torch.zeros(4, device=dev)
g = torch.Generator(device='cpu').manual_seed(7)
data = {'zeros': torch.zeros(9 * 64 * 8, dtype=torch.float16, device=dev),
        'random': (torch.randn(9 * 64 * 8, generator=g) * 0.5).half().to(dev)}
def ordered(order):
    return lambda f, sk, it, fl, st: lib.oake_debug_mfma_probe_order(f, sk, it, order, fl, st)
shapes = {'16x16x32': (lib.oake_debug_mfma_probe, 4000), '32x32x16': (lib.oake_debug_mfma_probe_32x32, 4000),
          '16x16x32 10x4 rows': (ordered(0), 2000), '16x16x32 10x4 serpentine': (ordered(1), 2000),
          '16x16x32 10x4 columns': (ordered(2), 2000), '16x16x32 10x4 col-serpentine': (ordered(3), 2000)}
for rnd in range(rounds):
    for dname, f in data.items():
        for sname, (fn, iters) in shapes.items():
            flop = C.c_double(0)
            def burst():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    rc = fn(f.data_ptr(), sink.data_ptr(), iters, C.byref(flop), s)
                    assert rc == 0, rc
                e1.record(); torch.cuda.synchronize()
                return flop.value * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12
            burst(); t0 = time.time(); series = []
            while time.time() - t0 < secs:
                series.append(burst())
            import statistics
            print(f'round {rnd} {dname:6s} {sname:30s}: first {series[0]:7.1f} ... last three {" ".join(f"{x:7.1f}" for x in series[-3:])}  mean {statistics.mean(series[1:]):7.1f} TFLOP/s', flush=True)
