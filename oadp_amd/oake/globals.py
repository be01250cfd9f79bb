"""``oadp.oake.globals``: one whole-image crop per image -> Tensor[512] f16.
Reference: oadp/oake/globals.py (Dataset :24-33, Validator :36-60)."""
from __future__ import annotations

import pathlib
from typing import NamedTuple

import PIL.Image
import torch

from .. import clip
from ..config import Config
from .base import BaseDataset, BaseValidator


class Batch(NamedTuple):
    output: pathlib.Path
    image: torch.Tensor


class Dataset(BaseDataset[Batch]):

    def _preprocess(self, id_: int, output: pathlib.Path, image: PIL.Image.Image) -> Batch:
        image = self.transforms.transform(image)
        return Batch(output, image)


class Validator(BaseValidator[Batch]):

    def _build_dataloader(self, config: Config):
        config = Config(config)
        config['dataset'] = Dataset(**config['dataset'])
        return super()._build_dataloader(config)

    @classmethod
    def _build_model(cls):
        return clip.load_default(True)

    def _encode(self, batches: list[Batch]) -> list[torch.Tensor]:
        # reference _run_iter (globals.py:49-60): encode_image -> F.normalize -> squeeze -> .half(),
        # here for a whole batch of images with normalise + fp16 cast fused into the head kernel
        images = torch.stack([b.image for b in batches]).to(self._device, non_blocking=True)
        emb = self._model.encode_image(images, normalize=True, out_dtype=torch.float16).cpu()
        return [emb[i].clone() for i in range(len(batches))]


if __name__ == '__main__':
    Validator.main()
