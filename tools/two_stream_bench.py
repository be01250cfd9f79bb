"""Experiment: batches of 256 crops alternating over 1, 2 or 3 handles / HIP streams (the kernels of
independent batches fill each other's start-up and tail) vs one stream — GPU box only.
Also the older question: one batch split in halves over two streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import clip
from oadp_amd.weights import synthetic_state_dict

dev = torch.device('cuda:0')
sd = synthetic_state_dict()
B = 256
x = torch.randn(B, 3, 224, 224, device=dev)


def run(nstream, batch, steps=60):
    models = [clip.load(sd, max_batch=batch)[0] for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    xs = x[:batch].contiguous()
    def step(i):
        with torch.cuda.stream(streams[i % nstream]):
            models[i % nstream].encode_image(xs, normalize=True, out_dtype=torch.float16)
    for i in range(3 * nstream):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{nstream} stream(s), batches of {batch}: {dt*1e3:.3f} ms per batch  {batch/dt:.0f} img/s', flush=True)
    del models


for n, b in ((1, 256), (2, 256), (3, 256), (1, 256), (2, 256), (2, 128)):
    run(n, b)
