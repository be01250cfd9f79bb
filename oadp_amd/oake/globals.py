"""``oadp.oake.globals``: one whole-image crop per image -> Tensor[512] f16.
Reference: oadp/oake/globals.py (Dataset :24-33, Validator :36-60)."""
from __future__ import annotations

import pathlib
from typing import NamedTuple

import PIL.Image
import torch

from .. import clip
from ..config import Config
from .base import BaseDataset, BaseValidator, image_to_u8


class Batch(NamedTuple):
    output: pathlib.Path
    image: torch.Tensor


class Dataset(BaseDataset[Batch]):

    def _preprocess(self, id_: int, output: pathlib.Path, image: PIL.Image.Image) -> Batch:
        if self._device_preprocess:
            return Batch(output, image_to_u8(image))  # resize / crop / normalise run on the GPU
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

    def _encode(self, batches: list[Batch]):
        # reference _run_iter (globals.py:49-60): encode_image -> F.normalize -> squeeze -> .half(),
        # here for a whole batch of images with normalise + fp16 cast fused into the head kernel
        if batches[0].image.dtype == torch.uint8:
            # device preprocessing: one uint8 HWC upload per image, Pillow-exact resize on the GPU
            squash = getattr(self._dataloader.dataset.transform, 'squash', False)
            decoded = self._images_u8([b.image for b in batches])
            images = self._model.visual.crop_resize_normalize_batch(
                decoded, [[(0, 0, im.shape[1], im.shape[0])] for im in decoded], squash=squash,
                out_dtype=torch.float16)
        else:
            images = self._to_device(torch.stack([b.image for b in batches]))
        host = self._to_host(self._model.encode_image(images, normalize=True, out_dtype=torch.float16))
        n = len(batches)

        def finish() -> list[torch.Tensor]:
            emb = host.get()
            return [emb[i].clone() for i in range(n)]

        return finish if images.is_cuda else finish()


if __name__ == '__main__':
    Validator.main()
