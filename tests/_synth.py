"""Synthetic COCO-style dataset + an oracle-backed stand-in model for CPU plumbing tests."""
import json
import pathlib
import pickle

import numpy as np
import PIL.Image
import torch

from oadp_amd.clip.preprocess import Preprocess
from oadp_amd.weights import synthetic_state_dict
from oracle.vit_ref import ViTConfig, encode_image_ref, encode_objects_ref

TINY = dict(width=128, layers=1, heads=2, mlp_dim=256, embed_dim=64)


def make_coco(root: pathlib.Path, sizes, proposals_per_image: int = 12, seed: int = 0, fmt: str = 'png'):
    """Writes <root>/images/*.jpg-like PNGs, an instances json (shuffled id order) and a proposals pkl
    aligned with SORTED ids (proposal_sorted=True)."""
    img_dir = root / 'images'
    img_dir.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    images = []
    ids = [int(i) for i in rng.permutation(np.arange(100, 100 + 7 * len(sizes), 7))[:len(sizes)]]
    for id_, (w, h) in zip(ids, sizes):
        arr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        name = f'{id_:012d}.{fmt}'
        if fmt == 'jpg':
            # smooth content + varied JPEG flavours (4:2:0 / 4:2:2 / 4:4:4, optimised tables, restarts,
            # progressive, and one CMYK file that the device decoder must hand back to PIL)
            yy, xx = np.mgrid[0:h, 0:w]
            arr = (arr // 4 + np.stack([(xx * 3 + yy) % 192, (xx + yy * 2) % 192, (xx * yy) % 192], -1)).astype(np.uint8)
            k = len(images)
            if k == 2:
                PIL.Image.fromarray(arr).convert('CMYK').save(img_dir / name, quality=85)
            else:
                PIL.Image.fromarray(arr).save(img_dir / name, quality=(95, 80, 60)[k % 3], subsampling=(2, 1, 0)[k % 3],
                                              optimize=k % 2 == 1, progressive=k in (1, 4),
                                              **(dict(restart_marker_blocks=7) if k == 3 else {}))
        else:
            PIL.Image.fromarray(arr).save(img_dir / name)
        images.append(dict(id=id_, file_name=name, width=w, height=h,
                           coco_url=f'http://images.cocodataset.org/images/{name}'))
    ann = root / 'instances.json'
    ann.write_text(json.dumps(dict(images=images, annotations=[], categories=[])))
    by_id = {im['id']: im for im in images}
    props = []
    for id_ in sorted(by_id):
        w, h = by_id[id_]['width'], by_id[id_]['height']
        n = proposals_per_image
        x1 = rng.uniform(0, w * 0.7, n)
        y1 = rng.uniform(0, h * 0.7, n)
        bw = rng.uniform(2, max(w * 0.5, 2.5), n)
        bh = rng.uniform(2, max(h * 0.5, 2.5), n)
        score = np.sort(rng.uniform(0, 1, n))[::-1]
        props.append(np.stack([x1, y1, np.minimum(x1 + bw, w), np.minimum(y1 + bh, h), score], 1).astype(np.float32))
    pkl = root / 'proposals.pkl'
    with open(pkl, 'wb') as f:
        pickle.dump(props, f)
    return dict(root=str(img_dir), annFile=str(ann), proposal_file=str(pkl), ids=sorted(by_id))


class _OracleVisual:
    """Test double with the model.visual surface, computing with the CPU oracle."""

    def __init__(self, sd):
        self._sd = dict(sd)
        self.grid = 7
        self.patch_size = 32
        self.object_stream = False
        self._cfg = ViTConfig(**TINY)

    def objects_mode(self):
        from oadp_amd.clip.model import VisionTransformer

        class _V:
            positional_embedding = self._sd['visual.positional_embedding']
        self._sd['visual.positional_embedding'] = VisionTransformer.interpolate_positional_embedding(_V, (14, 14))
        self.grid = 14
        self._cfg = ViTConfig(**TINY, stride=16, padding=15)
        self.object_stream = True

    def __call__(self, x, masks=None, *, normalize=False, out_dtype=None):
        if masks is None:
            out = encode_image_ref(self._sd, self._cfg, x.float().cpu())
        else:
            out = encode_objects_ref(self._sd, self._cfg, x.float().cpu(), masks.float().cpu())
        if normalize:
            out = torch.nn.functional.normalize(out)
        return out.to(out_dtype or torch.float16)


class OracleModel:
    dtype = torch.float16

    def __init__(self):
        self.visual = _OracleVisual(synthetic_state_dict(**TINY))

    def encode_image(self, x, **kw):
        return self.visual(x, **kw)


def preprocess():
    return Preprocess(224, squash=False)


def tiny_state_dict(**arch):
    """A small random-init vision state dict, stored the way OpenAI's ViT-B-32.pt stores its tensors:
    matmul / conv weights and the projection in fp16, LayerNorm parameters and embeddings in fp32."""
    sd = synthetic_state_dict(**arch)
    half = ('conv1.weight', 'in_proj_weight', 'in_proj_bias', 'out_proj.weight', 'out_proj.bias',
            'c_fc.weight', 'c_fc.bias', 'c_proj.weight', 'c_proj.bias', 'visual.proj')
    return {k: (v.half() if k.endswith(half) else v) for k, v in sd.items()}


def save_torchscript_checkpoint(sd, path) -> None:
    """``sd`` as a TorchScript archive whose ``torch.jit.load(path).state_dict()`` returns it — the
    container format of the reference's pretrained/clip/ViT-B-32.pt (README.md:129).  Includes the three
    scalar entries OpenAI's archives carry next to the weights."""
    import torch.nn as nn
    root = nn.Module()
    entries = dict(sd, input_resolution=torch.tensor(224), context_length=torch.tensor(77),
                   vocab_size=torch.tensor(49408))
    for key, value in entries.items():
        parts, m = key.split('.'), root
        for part in parts[:-1]:
            if part not in m._modules:
                m.add_module(part, nn.Module())
            m = m._modules[part]
        if value.is_floating_point():
            m.register_parameter(parts[-1], nn.Parameter(value.clone(), requires_grad=False))
        else:
            m.register_buffer(parts[-1], value.clone())
    torch.jit.save(torch.jit.script(root), str(path))
