"""Behaviours of the reference's un-vendored dependencies that nothing in /root/reference pins.

The arithmetic behind ``clip.load_default`` lives in the LutingWang/CLIP fork (reference README.md:44,
no pinned version) and the box helpers in ``todd`` (README.md:43); neither is vendored, and the
reference has no tests.  Three choices therefore rest on our reading of the call sites (SURVEY.md
Appendix D.1-D.3).  Each is a config key (``fork = dict(...)`` in configs/oake/base.py, applied by
``BaseValidator.main`` before the model is built) so that a maintainer who has the fork can flip it
without touching code:

  load_default_true        what ``clip.load_default(True)`` (globals mode, oadp/oake/globals.py:47) does to
                           the transform: 'squash' = Resize((n, n)) without cropping (default), or
                           'center_crop' = the OpenAI transform, same as ``load_default(False)``
  positional_interpolation ``visual.interpolate_positional_embedding`` (oadp/oake/objects.py:292-296):
                           mode + align_corners handed to F.interpolate; default bicubic / False
  min_wh_inclusive         ``todd.BBoxes.indices(min_wh=(4, 4))`` (oadp/oake/objects.py:163-165): True =
                           keep w >= 4 and h >= 4 (default), False = strictly greater
"""
from __future__ import annotations

_DEFAULTS = dict(load_default_true='squash',
                 positional_interpolation=dict(mode='bicubic', align_corners=False),
                 min_wh_inclusive=True)


class ForkSettings:

    def __init__(self) -> None:
        self.reset()

    def reset(self) -> None:
        self.load_default_true = _DEFAULTS['load_default_true']
        self.positional_interpolation = dict(_DEFAULTS['positional_interpolation'])
        self.min_wh_inclusive = _DEFAULTS['min_wh_inclusive']

    def configure(self, **kw) -> None:
        unknown = set(kw) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f'unknown fork setting(s) {sorted(unknown)}; known: {sorted(_DEFAULTS)}')
        if 'load_default_true' in kw:
            if kw['load_default_true'] not in ('squash', 'center_crop'):
                raise ValueError("load_default_true must be 'squash' or 'center_crop'")
            self.load_default_true = kw['load_default_true']
        if 'positional_interpolation' in kw:
            pi = dict(kw['positional_interpolation'])
            if set(pi) - {'mode', 'align_corners'} or pi.get('mode', 'bicubic') not in ('bicubic', 'bilinear', 'nearest'):
                raise ValueError('positional_interpolation = dict(mode=bicubic|bilinear|nearest, align_corners=bool)')
            self.positional_interpolation = dict(_DEFAULTS['positional_interpolation'], **pi)
        if 'min_wh_inclusive' in kw:
            self.min_wh_inclusive = bool(kw['min_wh_inclusive'])

    def as_dict(self) -> dict:
        return dict(load_default_true=self.load_default_true,
                    positional_interpolation=dict(self.positional_interpolation),
                    min_wh_inclusive=self.min_wh_inclusive)


settings = ForkSettings()
