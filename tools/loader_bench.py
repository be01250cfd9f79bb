"""Per-sample feature loading as the detector's dataloader does it (LoadCLIPFeatures: globals + blocks +
objects of one image): per-image .pth files vs the memory-mapped pack.  CPU only.
usage: loader_bench.py [n_images]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd.dp import LoadCLIPFeatures, pack
import pathlib
from oadp_amd.oake.base import atomic_save as _save
atomic_save = lambda obj, p: _save(obj, pathlib.Path(p))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
with tempfile.TemporaryDirectory() as root:
    g = torch.Generator().manual_seed(0)
    for mode in ('globals', 'blocks', 'objects'):
        os.makedirs(f'{root}/{mode}/train2017')
    for i in range(n):
        key = f'{i:012d}'
        atomic_save(torch.randn(1, 512, generator=g).half(), f'{root}/globals/train2017/{key}.pth')
        atomic_save(dict(embeddings=torch.randn(26, 512, generator=g).half(), bboxes=torch.rand(26, 4, generator=g).half() * 400),
                    f'{root}/blocks/train2017/{key}.pth')
        xy = torch.rand(300, 2, generator=g) * 400
        atomic_save(dict(embeddings=torch.randn(300, 512, generator=g).half(), bboxes=torch.cat([xy, xy + 50], 1).half(),
                         objectness=torch.rand(300, 1, generator=g).half()), f'{root}/objects/train2017/{key}.pth')
    t0 = time.perf_counter()
    for mode in ('globals', 'blocks', 'objects'):
        pack(f'{root}/{mode}', 'train2017')
    print(f'pack: {time.perf_counter() - t0:.2f} s for {n} images x 3 modes')
    for layer in ('PthAccessLayer', 'PackAccessLayer'):
        step = LoadCLIPFeatures(default=dict(task_name='train2017', type=layer),
                                globals_=dict(data_root=f'{root}/globals'), blocks=dict(data_root=f'{root}/blocks'),
                                objects=dict(data_root=f'{root}/objects'))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(n):
                step(dict(img_info=dict(id=i), bbox_fields=[]))
            best = min(best, time.perf_counter() - t0)
        print(f'{layer}: {best / n * 1e6:.0f} us per sample ({n / best:.0f} samples/s, page cache warm)')
