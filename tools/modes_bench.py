"""Throughput of the blocks / objects configurations of BASELINE.json (parity-test cases, not the
headline bench line): crops/s and images/s on one GPU, device-resident synthetic inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import clip
from oadp_amd.weights import synthetic_state_dict

dev = torch.device('cuda:0')
sd = synthetic_state_dict()

def timed(fn, steps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

# blocks: 64 images x 27 crops (640x480) = 1728 crops per step
model, _ = clip.load(sd, max_batch=512)
crops = torch.randn(1728, 3, 224, 224, device=dev, dtype=torch.float16)
t = timed(lambda: model.encode_image(crops, normalize=True, out_dtype=torch.float16))
print(f'blocks  (plain encode, 1728 crops = 64 images x 27): {t*1e3:.1f} ms  {1728/t:.0f} crops/s  {64/t:.0f} images/s')
del model

# objects: 300 proposals per image, mini-batch 512
model, _ = clip.load(sd, max_batch=512)
v = model.visual
v.positional_embedding = v.interpolate_positional_embedding((14, 14)); v.grid = 14
v.conv1.stride = (16, 16); v.conv1.padding = (15, 15); v.object_stream = True
objs = torch.randn(600, 3, 224, 224, device=dev, dtype=torch.float16)
masks = (torch.rand(600, 1, 14, 14, device=dev) > 0.5).half()
t = timed(lambda: v(objs, masks, normalize=True, out_dtype=torch.float16), steps=3, warm=1)
flop = 33_552_184_320
print(f'objects (dual-stream, 600 crops = 2 images x 300): {t*1e3:.1f} ms  {600/t:.0f} crops/s  {2/t:.1f} images/s  {600*flop/t/1e12:.0f} TFLOP/s (minimal-work)')
v.profile(True); v(objs, masks, normalize=True, out_dtype=torch.float16); prof = v.profile_read(); v.profile(False)
tot = sum(p['total_ms'] for p in prof)
for p in sorted(prof, key=lambda p: -p['total_ms'])[:12]:
    print(f"   {p['name']:20s} {p['total_ms']:8.3f} ms {100*p['total_ms']/tot:5.1f}%  {p['flops']/max(p['total_ms'],1e-9)/1e9:7.1f} TFLOP/s")
