"""Experiment: two half-batches on two streams (two handles) vs one full batch — GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import clip
from oadp_amd.weights import synthetic_state_dict

dev = torch.device('cuda:0')
sd = synthetic_state_dict()
B = 256
x = torch.randn(B, 3, 224, 224, device=dev)

def run(nsplit, steps=30):
    hb = B // nsplit
    models = [clip.load(sd, max_batch=hb)[0] for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    xs = [x[i * hb:(i + 1) * hb].contiguous() for i in range(nsplit)]
    def step():
        for m, s, xi in zip(models, streams, xs):
            with torch.cuda.stream(s):
                m.encode_image(xi, normalize=True, out_dtype=torch.float16)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{nsplit} stream(s) x {hb} crops: {dt*1e3:.3f} ms/step  {B/dt:.0f} img/s', flush=True)

for n in (1, 2, 4, 1, 2):
    run(n)
