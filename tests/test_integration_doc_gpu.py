"""INTEGRATION.md §B is code a maintainer of the reference would paste: run it as written.  The first python
block of that section (the ctypes binding of oake_create / oake_load_tensor / oake_encode_image) is executed
against a TorchScript checkpoint of the ViT-B/32 architecture — the container format of the reference's
``pretrained/clip/ViT-B-32.pt`` (README.md:129), here with synthetic weights — and its ``encode_image`` must
return, bit for bit, what the ``oadp_amd.clip`` facade returns for the same weights."""
import pathlib
import re

import pytest
import torch

from oadp_amd import _lib, clip
from oadp_amd.weights import synthetic_state_dict, normal

from . import _synth

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_ctypes_stub_of_integration_md_runs_as_written(cuda, tmp_path):
    text = (ROOT / 'INTEGRATION.md').read_text()
    section = text[text.index('## B. Bind the C ABI directly'):]
    block = re.search(r'```python\n(.*?)```', section, re.S).group(1)
    assert 'oake_encode_image' in block and "C.CDLL('oadp_amd/liboake_hip.so')" in block
    sd = synthetic_state_dict()
    ckpt = tmp_path / 'ViT-B-32.pt'
    _synth.save_torchscript_checkpoint(sd, ckpt)
    code = block.replace("'oadp_amd/liboake_hip.so'", repr(str(_lib.LIB_PATH)))
    code = code.replace("'pretrained/clip/ViT-B-32.pt'", repr(str(ckpt)))
    ns: dict = {}
    exec(compile(code, 'INTEGRATION.md#B', 'exec'), ns)  # defines lib, h, encode_image
    x = normal('doc_images', (5, 3, 224, 224), seed=11).to(cuda)
    got = ns['encode_image'](x)
    got16 = ns['encode_image'](x.half())
    model, _ = clip.load(sd, max_batch=256)
    ref = model.encode_image(x, normalize=True, out_dtype=torch.float16)
    assert got.shape == (5, 512) and got.dtype == torch.float16
    assert torch.equal(got, ref)
    assert torch.equal(got16, model.encode_image(x.half(), normalize=True, out_dtype=torch.float16))
    ns['lib'].oake_destroy.restype = None
    ns['lib'].oake_destroy.argtypes = [__import__('ctypes').c_void_p]
    ns['lib'].oake_destroy(ns['h'])
