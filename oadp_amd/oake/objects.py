"""``oadp.oake.objects``: proposal-driven square crops + 14x14 background masks, encoded with the
object-aware two-stream ViT -> dict(embeddings [N,512], bboxes [N,4], objectness [N,1]) f16.
Reference: oadp/oake/objects.py (COCODataset :43-195, Hooks :198-266, Validator :269-338).

The reference gets its box arithmetic from ``todd.BBoxes*`` (un-vendored).  Boxes here are plain
[n,4] float32 tensors (x1,y1,x2,y2); semantics per SURVEY.md §8c: wh = rb - lt, ``indices(min_wh)``
keeps w >= 4 and h >= 4 (>= vs > is unknown upstream — unpinned).
"""
from __future__ import annotations

import enum
import math
import os
import pathlib
import pickle
from typing import NamedTuple

import PIL.Image
import torch
import torch.nn.functional as F

from .. import clip
from ..config import Config
from ..store import Store
from .base import BaseDataset, BaseValidator, image_to_u8


class Batch(NamedTuple):
    output: pathlib.Path
    objects: torch.Tensor
    bboxes: torch.Tensor
    objectness: torch.Tensor
    masks: torch.Tensor
    crop_boxes: torch.Tensor | None = None  # device preprocessing: expanded boxes, `objects` = uint8 image


class ExpandMode(enum.Enum):
    RECTANGLE = enum.auto()
    LONGEST_EDGE = enum.auto()
    CONSTANT = enum.auto()
    ADAPTIVE = enum.auto()


def indices_min_wh(boxes: torch.Tensor, min_wh: tuple[float, float]) -> torch.Tensor:
    """``todd.BBoxes.indices(min_wh=...)``: >= by default, > with ``fork.min_wh_inclusive = False``
    (unknown upstream: oadp_amd/clip/settings.py)."""
    wh = boxes[:, 2:] - boxes[:, :2]
    if clip.settings.min_wh_inclusive:
        return (wh[:, 0] >= min_wh[0]) & (wh[:, 1] >= min_wh[1])
    return (wh[:, 0] > min_wh[0]) & (wh[:, 1] > min_wh[1])


class COCODataset(BaseDataset[Batch]):

    def __init__(self, *args, grid: int, expand_mode: str = 'ADAPTIVE', proposal_file: str,
                 proposal_sorted: bool, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._grid = grid
        self._expand_mode = ExpandMode[expand_mode]
        with open(proposal_file, 'rb') as f:
            proposals = pickle.load(f)
        # reference objects.py:68-74: proposals align with sorted ids, or annotation-file order
        ids = self.ids if proposal_sorted else list(self.coco.imgs.keys())
        self._proposals = {id_: torch.as_tensor(p, dtype=torch.float32).reshape(-1, 5)
                           for id_, p in zip(ids, proposals)}

    def _expand(self, bboxes: torch.Tensor, image_wh: torch.Tensor) -> torch.Tensor:
        """Square context boxes (reference objects.py:76-114).  ADAPTIVE: side = sqrt(8 * area),
        centred on the proposal, then shifted back inside the image where it sticks out; a box
        larger than the image stays centred."""
        wh = bboxes[:, 2:] - bboxes[:, :2]
        if self._expand_mode is ExpandMode.ADAPTIVE:
            length = torch.sqrt(wh[:, 0] * wh[:, 1] * 8).unsqueeze(1)
        elif self._expand_mode is ExpandMode.CONSTANT:
            length = torch.full((bboxes.shape[0], 1), 224.0)
        elif self._expand_mode is ExpandMode.LONGEST_EDGE:
            length = wh.max(dim=1, keepdim=True).values
        else:
            raise ValueError(self._expand_mode)
        center = (bboxes[:, :2] + bboxes[:, 2:]) / 2
        lt = center - length / 2
        rb = center + length / 2
        image_wh = image_wh.to(torch.float32)
        offset = torch.zeros_like(lt)
        offset = torch.where(lt >= 0, offset, -lt)
        offset = torch.where(rb <= image_wh, offset, image_wh - rb)
        offset = torch.where((rb - lt) <= image_wh, offset, torch.tensor(0.0))
        return torch.cat([lt + offset, rb + offset], dim=1)

    def _object(self, image: PIL.Image.Image, bbox) -> torch.Tensor:
        # PIL rounds each coordinate (banker's rounding) and zero-pads outside the image
        return self.transforms.transform(image.crop(tuple(float(v) for v in bbox)))

    def _mask(self, foreground, object) -> torch.Tensor:
        """[1,1,grid,grid]: 0 on the proposal, 1 on background (reference objects.py:129-155):
        a (y2-y1) x (x2-x1) pixel mask resampled 'nearest' to grid x grid."""
        x = torch.arange(object[2] - object[0])
        y = torch.arange(object[3] - object[1])
        inside_x = (foreground[0] <= x) & (x <= foreground[2])
        inside_y = (foreground[1] <= y) & (y <= foreground[3])
        mask = ~(inside_y[:, None] & inside_x[None, :])
        return F.interpolate(mask[None, None].float(), size=(self._grid, self._grid), mode='nearest')

    def _masks(self, foregrounds: torch.Tensor, objects: torch.Tensor) -> torch.Tensor:
        """``torch.cat([self._mask(fg, box) ...])`` for all proposals of an image at once.  'nearest'
        resampling is separable, so a mask is the outer product of 14 column tests and 14 row tests at
        the source indices torch picks (upsample_nearest: ``min(floor(dst * float(in) / out), in - 1)``,
        float32) — no per-proposal pixel masks.  Bit-identical to the per-proposal path
        (tests/test_pipeline_cpu.py), which cost 0.8 ms per proposal: 0.25 s per image of 300."""
        import numpy as np
        n, g = foregrounds.shape[0], self._grid
        if n == 0:
            return torch.zeros(0, 1, g, g)
        fg = foregrounds.to(torch.float32).numpy()
        box = objects.to(torch.float32).numpy().astype(np.float64)
        out = np.empty((n, 1, g, g), np.float32)
        dst = np.arange(g, dtype=np.float32)
        inside = []
        for lo, hi, a, b in ((0, 2, 0, 2), (1, 3, 1, 3)):
            length = np.ceil(box[:, hi] - box[:, lo]).astype(np.int64)  # len(torch.arange(x2 - x1))
            if (length < 1).any():
                raise ValueError('empty object box')
            scale = (length.astype(np.float32) / np.float32(g))[:, None]
            src = np.minimum(np.floor(dst[None, :] * scale).astype(np.int64), (length - 1)[:, None])
            src = src.astype(np.float32)
            inside.append((fg[:, a, None] <= src) & (src <= fg[:, b, None]))
        out[:, 0] = ~(inside[1][:, :, None] & inside[0][:, None, :])
        return torch.from_numpy(out)

    def _preprocess(self, id_: int, output: pathlib.Path, image: PIL.Image.Image) -> Batch:
        prop = self._proposals[id_]
        proposals, objectness = prop[:, :4], prop[:, 4:]
        keep = indices_min_wh(proposals, (4, 4))
        if Store.DRY_RUN:
            keep[5:] = False
        proposals, objectness = proposals[keep], objectness[keep]

        bboxes = self._expand(proposals, torch.tensor(image.size))
        foregrounds = proposals - torch.cat([bboxes[:, :2], bboxes[:, :2]], dim=1)

        if self._device_preprocess:
            # masks (index math) here; the crops are cut + resized on the GPU from the uint8 image.
            # `objects` carries the image, `crop_boxes` the expanded boxes (PIL crop semantics).
            return Batch(output, image_to_u8(image), proposals, objectness,
                         self._masks(foregrounds, bboxes), bboxes)
        objects = [self._object(image, box) for box in bboxes.tolist()]
        masks = self._masks(foregrounds, bboxes)
        if not objects:
            s = self.transform.n_px if hasattr(self.transform, 'n_px') else 224
            return Batch(output, torch.zeros(0, 3, s, s), proposals, objectness, masks)
        return Batch(output, torch.stack(objects), proposals, objectness, masks)


class LVISDataset(COCODataset):
    """LVIS annotations name their images by ``coco_url`` (train2017/... or val2017/... under the COCO root),
    reference objects.py:190-196; decoding — host PIL or device — is the base class's."""

    def _image_path(self, id_: int) -> str:
        info = self.coco.loadImgs([id_])[0]
        return os.path.join(self.root, info['coco_url'].replace('http://images.cocodataset.org/', ''))


DATASETS = dict(COCODataset=COCODataset, LVISDataset=LVISDataset)


class Validator(BaseValidator[Batch]):

    def __init__(self, *args, mini_batch_size: int, **kwargs) -> None:
        self._mini_batch_size = mini_batch_size
        super().__init__(*args, **kwargs)

    def _build_dataloader(self, config: Config):
        config = Config(config)
        ds = dict(config['dataset'])
        cls = DATASETS[ds.pop('type')]
        ds.setdefault('grid', self._model.visual.grid)
        config['dataset'] = cls(**ds)
        return super()._build_dataloader(config)

    @classmethod
    def _build_model(cls, upsample: int = 2):
        """The reference's surgery (objects.py:285-314) on our model facade: interpolated positional
        embedding, conv1 stride // upsample, padding (patch-1)//2, and the object-token stream
        (implemented in the HIP library instead of forward hooks)."""
        model, preprocess = clip.load_default(False)
        visual = model.visual
        visual.positional_embedding = visual.interpolate_positional_embedding((visual.grid * 2,) * 2)
        visual.grid *= upsample
        conv1 = visual.conv1
        conv1.stride = tuple(s // upsample for s in conv1.stride)
        conv1.padding = ((visual.patch_size - 1) // 2,) * 2
        visual.object_stream = True
        return model, preprocess

    def _n_crops(self, batch: Batch) -> int:
        return batch.bboxes.shape[0]

    def _encode(self, batches: list[Batch]):
        # reference _run_iter (objects.py:316-338): mini-batches of `mini_batch_size` crops through
        # model.visual(objects, masks), normalise, cat, .half() x3
        if batches[0].crop_boxes is not None:
            # device preprocessing: preprocess(image.crop(box)) for all proposals of all images of the
            # flush in three kernel launches, bit-exact with the PIL path
            objects = self._model.visual.crop_resize_normalize_batch(
                self._images_u8([b.objects for b in batches]), [b.crop_boxes for b in batches],
                out_dtype=torch.float16)
        else:
            objects = torch.cat([b.objects for b in batches])
        # the 0/1 masks of the whole flush go up once, as fp16 through a pinned slot (exact; the native
        # attention bias is -100 * mask either way)
        masks = self._to_device(torch.cat([b.masks for b in batches]).half())
        if not objects.is_cuda:
            objects = self._to_device(objects)
        embs = []
        # On the GPU the flush goes down in ONE call: the library cuts it into equal encoder passes of at most
        # min(mini_batch_size, clip.load's max_batch, ~25.6 k token rows / OAKE_PASS_ROWS) crops — `mini_batch_size` is
        # a memory bound in the reference and stays one here (visual.pass_limit -> OAKE_OPT_PASS_CROPS); a crop's
        # embedding depends on its pass only through the rounding of the last layer's object-token GEMMs
        # (tests/test_encoder_gpu.py::test_pass_cap_and_equal_passes_are_invisible).
        visual = self._model.visual
        if objects.is_cuda and hasattr(type(visual), 'pass_limit') and visual.pass_limit != self._mini_batch_size:
            visual.pass_limit = self._mini_batch_size
        step = objects.shape[0] if objects.is_cuda and objects.shape[0] else self._mini_batch_size
        for i in range(math.ceil(objects.shape[0] / step)):
            sl = slice(i * step, (i + 1) * step)
            embs.append(self._model.visual(objects[sl], masks[sl], normalize=True, out_dtype=torch.float16))
        on_gpu = bool(embs) and embs[0].is_cuda
        if on_gpu and not getattr(self, '_pass_logged', False) and hasattr(visual, 'get_option'):
            self._pass_logged = True  # the effective pass size decides the output's last-bit rounding: say what it is
            print(f'[{self.name}] encoder passes of at most {visual.get_option("pass_crops")} crops '
                  f'(mini_batch_size {self._mini_batch_size}, max_batch {getattr(visual, "max_batch", None)}, '
                  f'OAKE_PASS_ROWS {os.environ.get("OAKE_PASS_ROWS", "25600 (default)")}); flush of {objects.shape[0]} crops '
                  f'in one native call', flush=True)
        # one device -> host copy per flush, left in flight while the next flush is prepared (base._flush)
        host = self._to_host(torch.cat(embs)) if embs else None
        meta = [(b.bboxes.shape[0], b.bboxes.half(), b.objectness.half()) for b in batches]
        embed_dim = getattr(self._model.visual, 'output_dim', 512)  # (a flush without a single crop)

        def finish() -> list[dict]:
            emb = host.get() if host is not None else torch.zeros(0, embed_dim, dtype=torch.float16)
            out, i = [], 0
            for n, bboxes, objectness in meta:
                out.append(dict(embeddings=emb[i:i + n].clone(), bboxes=bboxes, objectness=objectness))
                i += n
            return out

        return finish if on_gpu else finish()


if __name__ == '__main__':
    Validator.main()
