"""``oadp.oake.blocks``: image pyramid tiled into 224x224 blocks (+ the whole-image crop) ->
dict(embeddings [K,512] f16, bboxes [K,4] f16).  Reference: oadp/oake/blocks.py."""
from __future__ import annotations

import itertools
import os
import pathlib
from typing import Generator, NamedTuple

import PIL.Image
import torch

from .. import clip
from ..config import Config
from .base import BaseDataset, BaseValidator, image_to_u8


class Batch(NamedTuple):
    output: pathlib.Path
    blocks: torch.Tensor
    bboxes: torch.Tensor


class Dataset(BaseDataset[Batch]):

    def __init__(self, *args, block_size: int = 224, max_stride: int = 112, rescale: float = 1.5,
                 **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._r = block_size
        self._s = max_stride
        self._rescale = rescale

    def _partition(self, length: int) -> list[int]:
        """Tile origins along one axis (reference blocks.py:40-52): n = ceil((len - r) / s) steps of
        near-equal integer size, the first ``rem`` steps one pixel longer."""
        if length < self._r:
            return []
        if length == self._r:
            return [0]
        n = (length - self._r - 1) // self._s + 1
        q, rem = divmod(length - self._r, n)
        origins = [0]
        for i in range(n):
            origins.append(origins[-1] + q + (1 if i < rem else 0))
        return origins

    def _partitions(self, image: PIL.Image.Image,
                    ) -> Generator[tuple[PIL.Image.Image, float, int, int], None, None]:
        """Pyramid walk (reference blocks.py:54-77): x-major product of the two partitions per
        level; next level = PIL bicubic resize to (int(w / 1.5), int(h / 1.5))."""
        scale = 1.0
        while True:
            w, h = image.size
            tiles = list(itertools.product(self._partition(w), self._partition(h)))
            if not tiles:
                return
            for x, y in tiles:
                yield image, scale, x, y
            image = image.resize((int(w / self._rescale), int(h / self._rescale)))
            scale *= self._rescale

    def _level_tiles(self, w: int, h: int) -> list[tuple[int, int, float, int, int]]:
        """Size-only twin of ``_partitions``: (level_w, level_h, scale, x, y) in the same order."""
        out, scale = [], 1.0
        while True:
            tiles = list(itertools.product(self._partition(w), self._partition(h)))
            if not tiles:
                return out
            out.extend((w, h, scale, x, y) for x, y in tiles)
            w, h = int(w / self._rescale), int(h / self._rescale)
            scale *= self._rescale

    def _block(self, image: PIL.Image.Image, x: int, y: int) -> torch.Tensor:
        return self.transforms.transform(image.crop((x, y, x + self._r, y + self._r)))

    def _bbox(self, scale: float, x: int, y: int) -> tuple[float, float, float, float]:
        x1, y1, r = x * scale, y * scale, self._r * scale
        return (x1, y1, x1 + r, y1 + r)

    def _preprocess(self, id_: int, output: pathlib.Path, image: PIL.Image.Image) -> Batch:
        # reference blocks.py:89-109.  Block 0 = whole image; its bbox is (x, y, side, side) —
        # NOT xyxy — a quirk of the reference that the consumer inherits, reproduced verbatim.
        w, h = image.size
        if self._device_preprocess:
            # index math only; the pyramid and the crops are produced on the GPU (Validator._encode)
            bboxes = [((w - h) / 2, 0, h, h) if w > h else (0, (h - w) / 2, w, w)]
            for _, _, scale, x, y in self._level_tiles(w, h):
                bboxes.append(self._bbox(scale, x, y))
            return Batch(output, image_to_u8(image), torch.tensor(bboxes))
        blocks = [self.transforms.transform(image)]
        bboxes: list[tuple] = [((w - h) / 2, 0, h, h) if w > h else (0, (h - w) / 2, w, w)]
        for level, scale, x, y in self._partitions(image):
            blocks.append(self._block(level, x, y))
            bboxes.append(self._bbox(scale, x, y))
        return Batch(output, torch.stack(blocks), torch.tensor(bboxes))


class Validator(BaseValidator[Batch]):

    def _build_dataloader(self, config: Config):
        config = Config(config)
        config['dataset'] = Dataset(**config['dataset'])
        return super()._build_dataloader(config)

    @classmethod
    def _build_model(cls):
        return clip.load_default(False)

    def _n_crops(self, batch: Batch) -> int:
        return batch.bboxes.shape[0]

    def _device_blocks(self, image_u8: torch.Tensor, out: torch.Tensor) -> None:
        """Blocks of one image on the GPU, written into ``out`` [k,3,r,r]: block 0 =
        preprocess(whole image); then per pyramid level exact 224x224 crops of the level image,
        levels chained by Pillow-exact resizes."""
        ds = self._dataloader.dataset
        v = self._model.visual
        level = image_u8
        h, w = level.shape[:2]
        v.crop_resize_normalize(level, [(0, 0, w, h)], out_dtype=torch.float16, out=out[0:1])
        r, i = ds._r, 1
        while True:
            tiles = list(itertools.product(ds._partition(w), ds._partition(h)))
            if not tiles:
                break
            v.crop_normalize(level, [(x, y, x + r, y + r) for x, y in tiles], out_dtype=torch.float16,
                             out=out[i:i + len(tiles)])
            i += len(tiles)
            w, h = int(w / ds._rescale), int(h / ds._rescale)
            level = v.resize_u8(level, (w, h))
        if i != out.shape[0]:
            raise RuntimeError(f'{i} blocks cut, {out.shape[0]} expected from the dataset\'s bboxes')

    def _encode(self, batches: list[Batch]):
        # reference _run_iter (blocks.py:125-135), crops of several images in one encoder pass
        if batches[0].blocks.dtype == torch.uint8:
            # device preprocessing: pyramids + every block crop of the whole flush in one native call
            # (oake_blocks_batch; `_device_blocks` is the same thing image by image, kept as its test twin)
            ds = self._dataloader.dataset
            counts = [b.bboxes.shape[0] for b in batches]
            images = self._images_u8([b.blocks for b in batches])
            if os.environ.get('OAKE_BLOCKS_PER_IMAGE'):  # A/B switch: the image-by-image composition
                blocks = torch.empty((sum(counts), 3, ds._r, ds._r), dtype=torch.float16, device=self._device)
                i = 0
                for im, k in zip(images, counts):
                    self._device_blocks(im, blocks[i:i + k])
                    i += k
            else:
                blocks, got = self._model.visual.blocks_batch(images, block_size=ds._r, max_stride=ds._s,
                                                              rescale=ds._rescale, out_dtype=torch.float16)
                if got != counts:
                    raise RuntimeError(f'{got} blocks cut, {counts} expected from the datasets\' bboxes')
        else:
            blocks = self._to_device(torch.cat([b.blocks for b in batches]))
            counts = [b.blocks.shape[0] for b in batches]
        host = self._to_host(self._model.encode_image(blocks, normalize=True, out_dtype=torch.float16))
        bboxes = [b.bboxes.half() for b in batches]

        def finish() -> list[dict]:
            emb = host.get()
            out, i = [], 0
            for bb, k in zip(bboxes, counts):
                out.append(dict(embeddings=emb[i:i + k].clone(), bboxes=bb))
                i += k
            return out

        return finish if blocks.is_cuda else finish()


if __name__ == '__main__':
    Validator.main()
