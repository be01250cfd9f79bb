"""How much of the GEMM rate is the power cap?  The globals bench with (a) the usual random weights and inputs,
(b) all-zero weights and inputs: the same instruction stream and memory traffic, but operand bits that do not
toggle.  GPU box only.  usage: zero_data_bench.py [steps=60]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else '60'
CODE = r'''
import sys, runpy, torch
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-profile', '--steps', '%s', '--warmup', '10']
if %d:
    import oadp_amd.weights as w
    real = w.synthetic_state_dict
    w.synthetic_state_dict = lambda *a, **k: {n: torch.zeros_like(t) for n, t in real(*a, **k).items()}
    _randn = torch.randn
    torch.randn = lambda *a, **k: _randn(*a, **k) * 0
runpy.run_path('%s/bench.py', run_name='__main__')
'''
for rnd in range(2):
    for zero in (0, 1):
        out = subprocess.run([sys.executable, '-c', CODE % (steps, zero, ROOT)], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        if not line:
            print(out.stdout[-2000:], out.stderr[-2000:]); continue
        d = json.loads(line[-1])
        k = d.get('kernels') or {}
        print('zeros ' if zero else 'random', d['value'], 'img/s;',
              {n: k[n]['tflops'] for n in ('gemm_c_fc', 'gemm_c_proj', 'gemm_qkv') if n in k}, flush=True)
