"""Which compute units does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) use on this board?
For a few masks: a census launch (one block per CU, each holding its CU 300 us) on the masked stream ->
distinct (XCC, SE, SH, CU) tuples seen, per-XCC counts.  Answers how mask bit i maps to (XCD, CU): what bench.py's
OAKE_BENCH_CU_SPLIT needs to give each lane half of EVERY XCD (so that block b still lands on XCD b % 8).
GPU box only."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oadp_amd import _lib
from oadp_amd.cumask import create_masked_stream, hip_runtime

lib = _lib.load()
dev = torch.device('cuda:0')
ncu = torch.cuda.get_device_properties(0).multi_processor_count
print('CUs', ncu)


def census(stream_ptr, nblocks=768, hold=300):
    out = torch.full((nblocks, 2), 0xFFFFFFFF, dtype=torch.int64, device=dev).to(torch.int32)
    torch.cuda.synchronize()
    rc = lib.oake_debug_cu_census(out.data_ptr(), nblocks, hold, C.c_void_p(stream_ptr))
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype('uint32')
    seen = collections.Counter()
    for xcc, hw in o:
        cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        seen[(int(xcc) & 15, int(se), int(sh), int(cu))] += 1
    per_xcc = collections.Counter(k[0] for k in seen)
    return seen, per_xcc


def mask_words(bits):
    words = [0] * ((ncu + 31) // 32)
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    return words


cases = {
    'no mask (stream 0)': None,
    'low half: bits 0..127': list(range(ncu // 2)),
    'high half: bits 128..255': list(range(ncu // 2, ncu)),
    'even bits': list(range(0, ncu, 2)),
    'bits with (i // 8) even': [i for i in range(ncu) if (i // 8) % 2 == 0],
    'bits with (i // 16) even': [i for i in range(ncu) if (i // 16) % 2 == 0],
    'bits 0..31': list(range(32)),
    'bits 0..7': list(range(8)),
}
for name, bits in cases.items():
    if bits is None:
        ptr = 0
    else:
        ptr = create_masked_stream(mask_words(bits))
    seen, per_xcc = census(ptr)
    print(f'{name:28s}: {len(seen):3d} distinct CUs; per XCC {dict(sorted(per_xcc.items()))}', flush=True)
    if bits is not None and len(bits) <= 32:
        print('      ', sorted(seen))
