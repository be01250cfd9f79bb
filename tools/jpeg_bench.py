"""Decode throughput: PIL (host, what the reference's workers do) vs host Huffman pass + GPU
reconstruction (oake_decode_jpeg) on a COCO-sized synthetic JPEG — GPU box only."""
import ctypes as C, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from PIL import Image
from oadp_amd import _lib, clip
from oadp_amd.weights import synthetic_state_dict
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))

lib = _lib.load()
model, _ = clip.load(synthetic_state_dict(width=64, layers=1, heads=1, mlp_dim=128, embed_dim=32), max_batch=2)
rng = np.random.default_rng(0)
for (h, w, q) in [(480, 640, 85), (427, 640, 75), (1134, 1700, 90)]:
    yy, xx = np.mgrid[0:h, 0:w]
    a = (rng.integers(0, 64, (h, w, 3)) + np.stack([(xx * 3 + yy) % 192, (xx + yy * 2) % 192, (xx * yy // 7) % 192], -1)).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(a).save(b, 'JPEG', quality=q, subsampling=2); data = b.getvalue()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): np.asarray(Image.open(io.BytesIO(data)).convert('RGB'))
    t_pil = (time.perf_counter() - t0) / n
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data); total = C.c_size_t(0)
    lib.oake_jpeg_entropy_decode(buf, len(data), None, 0, C.byref(total)); co = np.zeros(total.value, np.int16)
    t0 = time.perf_counter()
    for _ in range(n): lib.oake_jpeg_entropy_decode(buf, len(data), co.ctypes.data_as(C.c_void_p), co.size, C.byref(total))
    t_huff = (time.perf_counter() - t0) / n
    out = model.visual.decode_jpeg(data); torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.asarray(Image.open(io.BytesIO(data)).convert('RGB')))
    t0 = time.perf_counter()
    for _ in range(n): model.visual.decode_jpeg(data)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / n
    model.visual.profile(True)
    for _ in range(10): model.visual.decode_jpeg(data)
    p = [x for x in model.visual.profile_read() if x['name'] == 'jpeg_reconstruct'][0]; model.visual.profile(False)
    print(f'{w}x{h} q{q} ({len(data)/1024:.0f} KB): PIL {t_pil*1e3:.2f} ms | host Huffman {t_huff*1e3:.2f} ms | '
          f'oake_decode_jpeg end-to-end {t_dev*1e3:.2f} ms per image on one host thread | GPU kernels {p["total_ms"]/p["launches"]*1e3:.0f} us', flush=True)
