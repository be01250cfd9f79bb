"""Drop-in for the slice of the ``clip`` package (LutingWang/CLIP fork, reference README.md:44)
that OADP's OAKE path touches: ``load_default`` -> (model, preprocess), ``model.encode_image``,
``model.visual(objects, masks)``, ``model.dtype``, ``model.visual.grid``."""
from . import model
from .model import CLIP, VisionTransformer, load, load_default
from .preprocess import Preprocess
from .settings import settings

__all__ = ['model', 'CLIP', 'VisionTransformer', 'load', 'load_default', 'Preprocess', 'settings']
