"""``clip.model`` facade over liboake_hip.so.

Mirrors the attribute surface the reference touches:

    model, preprocess = clip.load_default(flag)          oadp/oake/globals.py:47
    model.encode_image(images) -> [N, 512]               oadp/oake/globals.py:57, blocks.py:129
    model.visual(objects, masks) -> [N, 512]             oadp/oake/objects.py:330
    model.dtype                                          oadp/oake/objects.py:328
    visual.grid, visual.patch_size, visual.conv1.stride/.padding,
    visual.positional_embedding, visual.interpolate_positional_embedding(size)
                                                         oadp/oake/objects.py:292-301

The forward itself runs in hand-written gfx950 kernels behind the C ABI; tensors must live on
the GPU.  There is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
from typing import Mapping

import torch
import torch.nn.functional as F

from .. import _lib
from .preprocess import Preprocess
from .settings import settings


def _pass_config(max_batch: int, env) -> tuple[int, int]:
    """(oake_config.max_batch, oake_config.pass_rows) from the caller's max_batch and the two experiment variables the
    LIBRARY used to read itself (the pass size decides output rounding, so it travels through the
    ABI's arguments): OAKE_PASS_ROWS = token rows per encoder pass (0 = no row cap; default: the library's 25 600),
    OAKE_PASS_CROPS = the cap in crops directly (then no row cap beside it)."""
    rows = env.get('OAKE_PASS_ROWS')
    crops = env.get('OAKE_PASS_CROPS')
    if crops not in (None, ''):
        return max(1, min(int(max_batch), int(crops))), -1
    if rows in (None, ''):
        return int(max_batch), 0
    return int(max_batch), (int(rows) if int(rows) > 0 else -1)


_TORCH2OAKE = {torch.float32: _lib.OAKE_F32, torch.float16: _lib.OAKE_F16,
               torch.bfloat16: _lib.OAKE_BF16}

VISION_PREFIX = 'visual.'


class Conv1Spec:
    """Stands in for ``visual.conv1`` (an nn.Conv2d in the reference): the reference's objects-mode
    surgery assigns ``conv1.stride`` and ``conv1.padding`` (objects.py:298-301)."""

    def __init__(self, patch_size: int) -> None:
        self.kernel_size = (patch_size, patch_size)
        self.stride = (patch_size, patch_size)
        self.padding = (0, 0)


def _infer_arch(sd: Mapping[str, torch.Tensor]) -> dict:
    conv = sd['visual.conv1.weight']
    width, patch = conv.shape[0], conv.shape[-1]
    pos = sd['visual.positional_embedding']
    grid = int(round((pos.shape[0] - 1) ** 0.5))
    layers = len({k.split('.')[3] for k in sd if k.startswith('visual.transformer.resblocks.')})
    mlp = sd['visual.transformer.resblocks.0.mlp.c_fc.weight'].shape[0]
    return dict(image_size=grid * patch, patch_size=patch, width=width, layers=layers,
                heads=width // 64, mlp_dim=mlp, embed_dim=sd['visual.proj'].shape[1])


class _RemovableHandle:
    """What ``nn.Module.register_forward_*hook`` returns: ``.remove()`` unregisters the hook."""

    def __init__(self, hooks: list, hook) -> None:
        self._hooks, self._hook = hooks, hook

    def remove(self) -> None:
        if self._hook in self._hooks:
            self._hooks.remove(self._hook)


class _HookPoint:
    """The forward-hook registration surface of an ``nn.Module``, recording only.

    The reference's objects mode attaches its ``Hooks`` object to ``visual``, ``visual.transformer``
    and every ``visual.transformer.resblocks[i]`` (oadp/oake/objects.py:303-312).  The native encoder
    has no Python-level module boundaries to call hooks at; what those particular hooks compute — the
    object-token stream — is implemented in the library (``oake_encode_objects``).  So hooks are
    recorded here, and ``VisionTransformer._hook_mode()`` recognises the reference's pattern (and
    refuses anything else at forward time, loudly)."""

    def __init__(self) -> None:
        self._forward_pre_hooks: list = []
        self._forward_hooks: list = []

    def register_forward_pre_hook(self, hook, **kwargs):
        self._forward_pre_hooks.append(hook)
        return _RemovableHandle(self._forward_pre_hooks, hook)

    def register_forward_hook(self, hook, **kwargs):
        self._forward_hooks.append(hook)
        return _RemovableHandle(self._forward_hooks, hook)


class _AttnSpec:
    """``resblock.attn``: the reference's hook reads ``module.attn.num_heads`` (objects.py:236)."""

    def __init__(self, embed_dim: int, num_heads: int) -> None:
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads


class ResidualAttentionBlock(_HookPoint):
    """``visual.transformer.resblocks[i]`` as far as the reference touches it: hook registration and
    ``attn.num_heads``.  The block's arithmetic runs in csrc/{gemm,attention}.hip."""

    def __init__(self, width: int, heads: int, index: int) -> None:
        super().__init__()
        self.attn = _AttnSpec(width, heads)
        self.index = index


class Transformer(_HookPoint):
    """``visual.transformer``: ``resblocks`` (iterable, indexable) + hook registration."""

    def __init__(self, width: int, layers: int, heads: int) -> None:
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = tuple(ResidualAttentionBlock(width, heads, i) for i in range(layers))


# method names of the reference's Hooks object, by the hook point they are registered on (objects.py:198-266)
_OBJECTS_HOOKS = dict(visual_pre='visual_forward_pre', transformer_pre='transformer_forward_pre',
                      transformer_post='transformer_forward', block_pre='residual_attention_block_forward_pre',
                      visual_post='visual_forward')


class VisionTransformer(_HookPoint):

    def __init__(self, state_dict: Mapping[str, torch.Tensor], *, compute_dtype=torch.float16,
                 residual_dtype: torch.dtype | None = None, max_batch: int = 512,
                 device: int | None = None, lib=None) -> None:
        # (max_batch: an upper bound on the crops per encoder pass — the library caps a pass at ~25.6 k token rows,
        # i.e. 512 crops at 50 tokens, 128 at 197 — csrc/api.hip oake_create)
        super().__init__()
        sd = {k: v.detach().to('cpu', torch.float32).contiguous()
              for k, v in state_dict.items() if k.startswith(VISION_PREFIX)}
        self._sd = sd
        arch = _infer_arch(sd)
        self.input_resolution = arch['image_size']
        self.patch_size = arch['patch_size']
        self.width = arch['width']
        self.layers = arch['layers']
        self.heads = arch['heads']
        self.mlp_dim = arch['mlp_dim']
        self.output_dim = arch['embed_dim']
        self.grid = self.input_resolution // self.patch_size
        self.conv1 = Conv1Spec(self.patch_size)
        self.transformer = Transformer(self.width, self.layers, self.heads)
        self.positional_embedding = sd['visual.positional_embedding']
        # The object-token stream of the reference's Hooks.  Switched on either explicitly (our mirror,
        # oadp_amd/oake/objects.py) or by registering the reference's own hook pattern (_hook_mode).
        self.object_stream = False
        self.compute_dtype = compute_dtype
        # residual stream x: the compute dtype (as the reference's fp16 GPU model) or float32
        self.residual_dtype = residual_dtype or compute_dtype
        self.max_batch = max_batch
        self.device = device
        # (lib: the variant tests and tools pass the lab build, which also carries the experiments)
        self._lib = lib if lib is not None else _lib.load()
        # Lanes: independent native handles (weights, activations and scratch each) so that consecutive
        # batches can run on different HIP streams and fill each other's kernel start-up and tail
        # (+10 % at batch 256, tools/two_stream_bench.py).  `lane` selects the handle every call on this
        # object uses; the caller pairs a lane with a stream (BaseValidator._flush, bench.py).
        self.lane = 0
        self._lanes: dict[int, tuple] = {}
        self._options: dict[int, int] = {}  # oake_set_option values, applied to every lane's handle
        # objects mode: crop_resize_normalize_batch writes its f16 crops straight into the zero-padded batch conv1 gathers
        # from (OAKE_LAYOUT_PADDED) and hands out a strided [N,3,S,S] VIEW of it; visual(objects, masks) recognises the view
        # and the library skips its pad pass.  One reusable pool per (lane, pool_slot): a view is valid until the next crop
        # call on the same lane and slot (the validators and bench.py encode a flush before they crop the next one).
        self.padded_crops = os.environ.get('OAKE_PADDED_CROPS', '1') not in ('0', 'false', 'False')
        self._pad_pools: dict[tuple, tuple] = {}

    @property
    def _handle(self):
        return self._lanes.get(self.lane, (None, None))[0]

    OPTIONS = dict(cls_last=_lib.OAKE_OPT_CLS_LAST, gemm_variant=_lib.OAKE_OPT_GEMM_VARIANT,
                   gemm_panel=_lib.OAKE_OPT_GEMM_PANEL, attention_variant=_lib.OAKE_OPT_ATTENTION_VARIANT,
                   patch_direct=_lib.OAKE_OPT_PATCH_DIRECT, cu_count=_lib.OAKE_OPT_CU_COUNT,
                   fuse_attn_out=_lib.OAKE_OPT_FUSE_ATTN_OUT, pass_crops=_lib.OAKE_OPT_PASS_CROPS,
                   fuse_qkv_attn=_lib.OAKE_OPT_FUSE_QKV_ATTN, qkv_walk=_lib.OAKE_OPT_QKV_WALK)

    def set_option(self, name: str, value: int) -> None:
        """Per-model kernel-selection switch (``oake_set_option`` on every lane's handle, now and for
        handles created later): cls_last, gemm_variant, gemm_panel, attention_variant."""
        key = self.OPTIONS[name]
        for h, _ in self._lanes.values():  # (a refused value raises here and is not remembered for later handles)
            _lib.check(self._lib, h, self._lib.oake_set_option(h, key, int(value)), 'oake_set_option')
        self._options[key] = int(value)

    # -- reference surface -----------------------------------------------------------------
    def interpolate_positional_embedding(self, size: tuple[int, int]) -> torch.Tensor:
        """[1 + g*g, C] -> [1 + size[0]*size[1], C]: CLS row kept, grid rows resampled — bicubically
        with align_corners=False unless ``fork.positional_interpolation`` says otherwise: the fork's own
        mode is unknown (oadp_amd/clip/settings.py)."""
        pos = self.positional_embedding.float()
        g = int(round((pos.shape[0] - 1) ** 0.5))
        cls, grid = pos[:1], pos[1:].reshape(1, g, g, -1).permute(0, 3, 1, 2)
        pi = settings.positional_interpolation
        if pi['mode'] == 'nearest':
            grid = F.interpolate(grid, size=size, mode='nearest')
        else:
            grid = F.interpolate(grid, size=size, mode=pi['mode'], align_corners=pi['align_corners'])
        grid = grid.permute(0, 2, 3, 1).reshape(size[0] * size[1], -1)
        return torch.cat([cls, grid])

    def _hook_mode(self) -> str:
        """'none' (no hooks anywhere), 'objects' (exactly the reference's objects-mode pattern,
        oadp/oake/objects.py:303-312: a pre-hook on visual, a pre- and a post-hook on transformer, a
        pre-hook on every block, all methods of one ``Hooks``-like object, identified by name) —
        anything else raises: arbitrary Python hooks cannot run inside the native encoder."""
        t = self.transformer
        points = [('visual_pre', self._forward_pre_hooks), ('visual_post', self._forward_hooks),
                  ('transformer_pre', t._forward_pre_hooks), ('transformer_post', t._forward_hooks)]
        points += [('block_pre', b._forward_pre_hooks) for b in t.resblocks]
        block_post = [h for b in t.resblocks for h in b._forward_hooks]
        if not block_post and not any(hooks for _, hooks in points):
            return 'none'

        def only(role, hooks, required=True):
            if not hooks and not required:
                return None
            if len(hooks) != 1 or getattr(hooks[0], '__name__', None) != _OBJECTS_HOOKS[role]:
                raise NotImplementedError(
                    f'forward hooks on the native encoder: only the OAKE objects-mode pattern '
                    f'(reference oadp/oake/objects.py:303-312) is supported; got '
                    f'{[getattr(h, "__name__", repr(h)) for h in hooks]} where {_OBJECTS_HOOKS[role]!r} belongs')
            return getattr(hooks[0], '__self__', None)

        if block_post:
            raise NotImplementedError('forward (post) hooks on resblocks are not part of the objects-mode pattern')
        owners = {id(only(role, hooks, required=(role != 'visual_post'))) for role, hooks in points
                  if not (role == 'visual_post' and not hooks)}
        if len(owners) != 1:
            raise NotImplementedError('the objects-mode hooks must be methods of ONE Hooks object')
        return 'objects'

    def _objects_mode(self) -> bool:
        return self.object_stream or self._hook_mode() == 'objects'

    def __call__(self, x: torch.Tensor, masks: torch.Tensor | None = None, *,
                 normalize: bool = False, out_dtype: torch.dtype | None = None) -> torch.Tensor:
        objects_mode = self._objects_mode()
        if masks is None:
            if objects_mode:
                raise ValueError('objects-mode model: call visual(objects, masks)')
            return self._forward(x, None, normalize, out_dtype)
        if not objects_mode:
            raise ValueError('visual(x, masks) needs the objects-mode surgery: the reference\'s '
                             'Validator._build_model (geometry + its Hooks registered on visual / '
                             'transformer / resblocks) or visual.object_stream = True')
        return self._forward(x, masks, normalize, out_dtype)

    forward = __call__

    # -- native handle ---------------------------------------------------------------------
    def _geometry(self) -> tuple[int, int]:
        stride = self.conv1.stride[0] if isinstance(self.conv1.stride, (tuple, list)) else self.conv1.stride
        pad = self.conv1.padding[0] if isinstance(self.conv1.padding, (tuple, list)) else self.conv1.padding
        return int(stride), int(pad)

    def _ensure_handle(self, device_index: int):
        stride, pad = self._geometry()
        pos = self.positional_embedding
        pos = pos.data if isinstance(pos, torch.nn.Parameter) else pos
        key = (device_index, stride, pad, pos.data_ptr(), tuple(pos.shape), self.compute_dtype,
               self.residual_dtype, self.max_batch, _pass_config(self.max_batch, os.environ))
        cur = self._lanes.get(self.lane)
        if cur is not None and key == cur[1]:
            return cur[0]
        if cur is not None:  # geometry / dtype changed (objects-mode surgery): rebuild this lane
            self._lib.oake_destroy(cur[0])
            del self._lanes[self.lane]
        lib = self._lib
        cfg = _lib.OakeConfig()
        lib.oake_default_config(C.byref(cfg))
        cfg.image_size, cfg.patch_size = self.input_resolution, self.patch_size
        cfg.stride, cfg.padding = stride, pad
        cfg.width, cfg.layers, cfg.heads = self.width, self.layers, self.heads
        cfg.mlp_dim, cfg.embed_dim = self.mlp_dim, self.output_dim
        cfg.compute_dtype = _TORCH2OAKE[self.compute_dtype]
        cfg.residual_dtype = _TORCH2OAKE[self.residual_dtype]
        cfg.max_batch, cfg.pass_rows = _pass_config(self.max_batch, os.environ)
        h = C.c_void_p()
        _lib.check(lib, None, lib.oake_create(C.byref(cfg), device_index, C.byref(h)), 'oake_create')
        try:
            tokens = lib.oake_tokens(h)
            if pos.shape[0] != tokens:
                raise ValueError(f'positional_embedding has {pos.shape[0]} rows, geometry needs {tokens}')
            for name, t in self._sd.items():
                if name == 'visual.positional_embedding':
                    t = pos.detach().to('cpu', torch.float32).contiguous()
                _lib.check(lib, h, lib.oake_load_tensor(h, name.encode(), t.data_ptr(), t.numel()),
                           f'oake_load_tensor({name})')
            missing = lib.oake_missing_tensors(h)
            if missing:
                raise ValueError(f'state_dict lacks {missing} vision-tower tensors')
            for opt, value in self._options.items():
                _lib.check(lib, h, lib.oake_set_option(h, opt, value), 'oake_set_option')
            self._apply_pass_limit(h)
        except Exception:
            lib.oake_destroy(h)
            raise
        self._lanes[self.lane] = (h, key)
        return h

    def get_option(self, name: str) -> int | None:
        """``oake_get_option`` on the current lane's handle (None before the first call created it)."""
        h = self._handle
        if h is None:
            return None
        v = C.c_int(0)
        _lib.check(self._lib, h, self._lib.oake_get_option(h, self.OPTIONS[name], C.byref(v)), 'oake_get_option')
        return v.value

    # crops per encoder pass: the reference's `mini_batch_size` [REF oadp/oake/objects.py:321-331] is a memory bound on
    # one pass; here a call's crops are cut into passes by the library (csrc/api.hip plan_pass_size: full passes + a shorter
    # one, or equal passes, whichever fills whole rounds of tiles; cap = min(max_batch, ~25.6 k token rows,
    # OAKE_PASS_ROWS) at handle creation) and `pass_limit` lowers that cap — it never raises it
    _pass_limit: int | None = None

    @property
    def pass_limit(self) -> int | None:
        return self._pass_limit

    @pass_limit.setter
    def pass_limit(self, n: int | None) -> None:
        self._pass_limit = None if n is None else max(1, int(n))
        for h, _ in self._lanes.values():
            self._apply_pass_limit(h)

    def _apply_pass_limit(self, h) -> None:
        # None: back to the cap the handle was created with (the library clamps the value to that cap), so a user of the
        # same model after an objects sweep — whose mini_batch_size lowered it — runs its own pass size again
        limit = self._pass_limit if self._pass_limit is not None else 2 ** 30
        _lib.check(self._lib, h, self._lib.oake_set_option(h, _lib.OAKE_OPT_PASS_CROPS, limit), 'oake_set_option')

    def close(self) -> None:
        for h, _ in self._lanes.values():
            self._lib.oake_destroy(h)
        self._lanes.clear()
        self._pad_pools.clear()

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _forward(self, x, masks, normalize, out_dtype) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError('oadp_amd.clip runs on the GPU only (no CPU fallback): move the '
                               'input to a HIP device')
        if x.dtype not in _TORCH2OAKE:
            raise TypeError(f'unsupported input dtype {x.dtype}')
        s = self.input_resolution
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, s, s):
            raise ValueError(f'expected [N,3,{s},{s}], got {tuple(x.shape)}')
        out_dtype = out_dtype or self.compute_dtype
        if out_dtype == torch.bfloat16:
            raw_dtype = torch.float32
        elif out_dtype in (torch.float32, torch.float16):
            raw_dtype = out_dtype
        else:
            raise TypeError(f'unsupported output dtype {out_dtype}')
        dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
        padded_base = self._padded_view_base(x) if masks is not None else None
        if padded_base is None:
            x = x.contiguous()
        n = x.shape[0]
        out = torch.empty((n, self.output_dim), dtype=raw_dtype, device=x.device)
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            lib = self._lib
            if masks is None:
                rc = lib.oake_encode_image(h, x.data_ptr(), _TORCH2OAKE[x.dtype], n, out.data_ptr(),
                                           _TORCH2OAKE[raw_dtype], int(normalize), stream)
                _lib.check(lib, h, rc, 'oake_encode_image')
            else:
                g = lib.oake_grid(h)
                if masks.dtype not in (torch.float32, torch.float16):
                    masks = masks.float()
                if masks.shape[0] != n or masks.numel() != n * g * g:
                    raise ValueError(f'masks must be [N,1,{g},{g}], got {tuple(masks.shape)}')
                masks = masks.to(x.device).contiguous()
                if padded_base is not None:  # a view of the zero-padded crop pool: no pad pass in the library
                    rc = lib.oake_encode_objects(h, padded_base, _TORCH2OAKE[x.dtype] | _lib.OAKE_LAYOUT_PADDED,
                                                 masks.data_ptr(), _TORCH2OAKE[masks.dtype], n, out.data_ptr(),
                                                 _TORCH2OAKE[raw_dtype], int(normalize), stream)
                else:
                    rc = lib.oake_encode_objects(h, x.data_ptr(), _TORCH2OAKE[x.dtype], masks.data_ptr(),
                                                 _TORCH2OAKE[masks.dtype], n, out.data_ptr(),
                                                 _TORCH2OAKE[raw_dtype], int(normalize), stream)
                _lib.check(lib, h, rc, 'oake_encode_objects')
        return out if out.dtype == out_dtype else out.to(out_dtype)

    # -- device-side preprocessing (csrc/resample.hip, rowops.hip) -------------------------------
    def _padded_layout(self, h) -> tuple[int, int, int] | None:
        """(padding, rows, row_stride) of the zero-padded batch this handle's conv1 gathers from, or None."""
        pad, hp, ws = C.c_int(0), C.c_int(0), C.c_int(0)
        if self._lib.oake_padded_layout(h, C.byref(pad), C.byref(hp), C.byref(ws)) != _lib.OAKE_OK:
            return None
        return pad.value, hp.value, ws.value

    def _padded_view_base(self, x: torch.Tensor) -> int | None:
        """Device address of plane 0 / row 0 of the first crop if `x` is a [k,3,S,S] view of one of this model's padded
        crop pools (a dim-0 slice of what crop_resize_normalize_batch returned), else None."""
        if x.is_contiguous() or not self._pad_pools or x.dim() != 4:
            return None
        for pool, pad, hp, ws in self._pad_pools.values():
            plane = hp * ws
            if (x.dtype == pool.dtype and x.device == pool.device and x.stride() == (3 * plane, plane, ws, 1)
                    and x.untyped_storage().data_ptr() == pool.untyped_storage().data_ptr()
                    and x.storage_offset() % (3 * plane) == pad * ws + pad):
                return x.data_ptr() - (pad * ws + pad) * x.element_size()
        return None

    def _padded_pool(self, h, k: int, slot: int, device: torch.device):
        """The zero-filled pool [cap >= k, 3, rows, row_stride] of (lane, slot) for this handle's geometry, or None where the
        handle takes no padded batch.  Created (and grown) with zeros; the writer keeps every border element zero."""
        lay = self._padded_layout(h)
        if lay is None:
            return None
        pad, hp, ws = lay
        key = (self.lane, slot)
        cur = self._pad_pools.get(key)
        if cur is None or cur[0].shape[0] < k or cur[1:] != (pad, hp, ws) or cur[0].device != device:
            cap = max(k, cur[0].shape[0] if cur is not None and cur[1:] == (pad, hp, ws) else 0)
            cur = (torch.zeros((cap, 3, hp, ws), dtype=torch.float16, device=device), pad, hp, ws)
            self._pad_pools[key] = cur
        return cur

    def _image_args(self, image_u8: torch.Tensor):
        if not image_u8.is_cuda or image_u8.dtype != torch.uint8 or image_u8.dim() != 3 or image_u8.shape[2] != 3:
            raise ValueError('expected a uint8 HWC RGB tensor on the GPU')
        dev = image_u8.device.index if image_u8.device.index is not None else torch.cuda.current_device()
        return image_u8.contiguous(), dev

    def _crop_out(self, out: torch.Tensor | None, k: int, n: int, out_dtype: torch.dtype,
                  device: torch.device) -> torch.Tensor:
        """The [k,3,n,n] destination of a crop call: freshly allocated, or the caller's slice of a
        larger batch tensor (a sweep fills one tensor per flush instead of concatenating hundreds)."""
        if out is None:
            return torch.empty((k, 3, n, n), dtype=out_dtype, device=device)
        if (tuple(out.shape) != (k, 3, n, n) or out.dtype != out_dtype or out.device != device
                or not out.is_contiguous()):
            raise ValueError(f'out must be a contiguous {out_dtype} tensor of shape {(k, 3, n, n)} on {device}')
        return out

    def crop_resize_normalize(self, image_u8: torch.Tensor, boxes, *, squash: bool = False,
                              out_dtype: torch.dtype = torch.float32,
                              out: torch.Tensor | None = None) -> torch.Tensor:
        """``torch.stack([preprocess(image.crop(box)) for box in boxes])`` on the device, bit-exact
        with Pillow + torchvision (Image.crop, Resize(n, BICUBIC), CenterCrop, ToTensor, Normalize).
        ``boxes``: [k,4] float (x1,y1,x2,y2), any device."""
        from .preprocess import CLIP_MEAN, CLIP_STD
        image_u8, dev = self._image_args(image_u8)
        boxes = torch.as_tensor(boxes, dtype=torch.float32).reshape(-1, 4).cpu().contiguous()
        k, n = boxes.shape[0], self.input_resolution
        out = self._crop_out(out, k, n, out_dtype, image_u8.device)
        if k == 0:
            return out
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            mean, std = (C.c_float * 3)(*CLIP_MEAN), (C.c_float * 3)(*CLIP_STD)
            rc = self._lib.oake_crop_resize_normalize(
                h, image_u8.data_ptr(), image_u8.shape[0], image_u8.shape[1],
                C.cast(boxes.data_ptr(), C.POINTER(C.c_float)), k, n, int(squash), mean, std,
                out.data_ptr(), _TORCH2OAKE[out_dtype], C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_crop_resize_normalize')
        return out

    def crop_resize_normalize_batch(self, images_u8: list[torch.Tensor], boxes: list, *, squash: bool = False,
                                    out_dtype: torch.dtype = torch.float32, pool_slot: int = 0) -> torch.Tensor:
        """``torch.cat([crop_resize_normalize(im, b) for im, b in zip(images_u8, boxes)])`` in one native
        call: ``boxes[i]`` ([k_i,4] floats) are the crops of ``images_u8[i]``.

        On an objects-mode model (conv1 with padding) with f16 output the result is a strided VIEW of the zero-padded crop
        pool of (lane, pool_slot) — same shape, same values, valid until the next call on that lane and slot; pass it (or
        dim-0 slices of it) to ``visual(objects, masks)`` and the encoder reads the pool in place.  ``.contiguous()`` gives
        an independent dense copy; ``padded_crops = False`` / OAKE_PADDED_CROPS=0 switches the pool off."""
        from .preprocess import CLIP_MEAN, CLIP_STD
        if len(images_u8) != len(boxes):
            raise ValueError('one box list per image')
        n = self.input_resolution
        if not images_u8:
            return torch.empty((0, 3, n, n), dtype=out_dtype, device='cuda')
        imgs, dev = [], None
        for im in images_u8:
            im, d = self._image_args(im)
            if dev is not None and d != dev:
                raise ValueError('all images must be on one device')
            imgs.append(im)
            dev = d
        bs = [torch.as_tensor(b, dtype=torch.float32).reshape(-1, 4) for b in boxes]
        counts = [b.shape[0] for b in bs]
        allb = torch.cat(bs).cpu().contiguous()
        total = sum(counts)
        m = len(imgs)
        ptrs = (C.c_void_p * m)(*[im.data_ptr() for im in imgs])
        hs = (C.c_int * m)(*[im.shape[0] for im in imgs])
        ws = (C.c_int * m)(*[im.shape[1] for im in imgs])
        cs = (C.c_int * m)(*counts)
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            pool = None
            if (self.padded_crops and total > 0 and out_dtype == torch.float16 and self.compute_dtype == torch.float16
                    and self.residual_dtype == torch.float16):
                pool = self._padded_pool(h, total, pool_slot, imgs[0].device)
            if pool is not None:
                buf, pad, _, _ = pool
                out, dst, flag = buf[:total, :, pad:pad + n, pad:pad + n], buf.data_ptr(), _lib.OAKE_LAYOUT_PADDED
            else:
                out = torch.empty((total, 3, n, n), dtype=out_dtype, device=imgs[0].device)
                dst, flag = out.data_ptr(), 0
            mean, std = (C.c_float * 3)(*CLIP_MEAN), (C.c_float * 3)(*CLIP_STD)
            rc = self._lib.oake_crop_resize_normalize_batch(
                h, m, ptrs, hs, ws, C.c_void_p(allb.data_ptr()), cs, n, int(squash), mean, std, dst,
                _TORCH2OAKE[out_dtype] | flag, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_crop_resize_normalize_batch')
        return out

    def blocks_count(self, width: int, height: int, block_size: int = 224, max_stride: int = 112,
                     rescale: float = 1.5) -> int:
        n = self._lib.oake_blocks_count(int(width), int(height), int(block_size), int(max_stride), float(rescale))
        if n < 0:
            raise ValueError('bad block geometry')
        return n

    def blocks_batch(self, images_u8: list[torch.Tensor], *, block_size: int = 224, max_stride: int = 112,
                     rescale: float = 1.5, out_dtype: torch.dtype = torch.float16,
                     out: torch.Tensor | None = None) -> tuple[torch.Tensor, list[int]]:
        """What the reference's blocks ``Dataset._preprocess`` (oadp/oake/blocks.py:89-109) produces for
        every image of a flush — block 0 = preprocess(whole image), then the 224x224 blocks of every level
        of the rescale pyramid — in ONE native call (``oake_blocks_batch``): bit-exact with the PIL path,
        about four launches per pyramid level for the whole flush.  Returns ([sum k_i, 3, r, r], [k_i])."""
        from .preprocess import CLIP_MEAN, CLIP_STD
        if not images_u8:
            return torch.empty((0, 3, block_size, block_size), dtype=out_dtype, device='cuda'), []
        imgs, dev = [], None
        for im in images_u8:
            im, d = self._image_args(im)
            if dev is not None and d != dev:
                raise ValueError('all images must be on one device')
            imgs.append(im)
            dev = d
        counts = [self.blocks_count(im.shape[1], im.shape[0], block_size, max_stride, rescale) for im in imgs]
        out = self._crop_out(out, sum(counts), block_size, out_dtype, imgs[0].device)
        m = len(imgs)
        ptrs = (C.c_void_p * m)(*[im.data_ptr() for im in imgs])
        hs = (C.c_int * m)(*[im.shape[0] for im in imgs])
        ws = (C.c_int * m)(*[im.shape[1] for im in imgs])
        got = (C.c_int * m)()
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            mean, std = (C.c_float * 3)(*CLIP_MEAN), (C.c_float * 3)(*CLIP_STD)
            rc = self._lib.oake_blocks_batch(h, m, ptrs, hs, ws, int(block_size), int(max_stride), float(rescale),
                                             mean, std, out.data_ptr(), _TORCH2OAKE[out_dtype], got,
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_blocks_batch')
        assert list(got) == counts
        return out, counts

    def resize_u8(self, image_u8: torch.Tensor, size: tuple[int, int]) -> torch.Tensor:
        """``PIL.Image.resize(size)`` (bicubic) of a uint8 HWC device image; ``size`` = (w, h)."""
        image_u8, dev = self._image_args(image_u8)
        w, h_ = int(size[0]), int(size[1])
        out = torch.empty((h_, w, 3), dtype=torch.uint8, device=image_u8.device)
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            rc = self._lib.oake_resize_u8(h, image_u8.data_ptr(), image_u8.shape[0], image_u8.shape[1],
                                          out.data_ptr(), h_, w,
                                          C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_resize_u8')
        return out

    def decode_jpeg(self, data: bytes, device: torch.device | None = None, *,
                    coefs: torch.Tensor | None = None) -> torch.Tensor:
        """Baseline JPEG file bytes -> uint8 HWC RGB device tensor, bit-identical to
        ``PIL.Image.open(...).convert('RGB')`` (the decode behind oadp/oake/base.py:53).  Huffman
        decoding runs on the calling thread, IDCT / upsampling / colour conversion on the GPU.
        ``coefs``: the int16 output of ``oake_jpeg_entropy_decode`` when the Huffman pass already ran
        elsewhere (a DataLoader worker) — then only the upload + GPU half runs here.
        Raises ``OakeError`` for files outside the supported subset (progressive, CMYK, ...)."""
        dev = torch.device(device).index if device is not None else None
        dev = torch.cuda.current_device() if dev is None else dev
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        hh, ww = C.c_int(0), C.c_int(0)
        rc = self._lib.oake_jpeg_info(buf, len(data), C.byref(hh), C.byref(ww), None)
        if rc != _lib.OAKE_OK:
            raise _lib.OakeError('oake_jpeg_info: ' + ('unsupported JPEG variant' if rc == _lib.OAKE_ERR_UNSUPPORTED
                                                       else 'not a valid JPEG'))
        out = torch.empty((hh.value, ww.value, 3), dtype=torch.uint8, device=torch.device('cuda', dev))
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if coefs is None:
                rc = self._lib.oake_decode_jpeg(h, buf, len(data), out.data_ptr(), out.numel(), C.byref(hh),
                                                C.byref(ww), stream)
            else:
                if coefs.dtype != torch.int16 or coefs.is_cuda or not coefs.is_contiguous():
                    raise ValueError('coefs must be a contiguous int16 CPU tensor')
                rc = self._lib.oake_jpeg_reconstruct(h, buf, len(data), C.c_void_p(coefs.data_ptr()),
                                                     coefs.numel(), out.data_ptr(), out.numel(),
                                                     C.byref(hh), C.byref(ww), stream)
            _lib.check(self._lib, h, rc, 'oake_decode_jpeg')
        return out

    def decode_jpeg_batch(self, datas: list, device: torch.device | None = None, *,
                          threads: int = 16) -> list[torch.Tensor | None]:
        """``decode_jpeg`` for many files in one native call: the Huffman passes run on ``threads``
        host threads inside the library (no GIL), the GPU half on the current stream.  ``datas``:
        ``bytes`` objects or 1-D uint8 CPU tensors (read in place, no copy).  The images are views of
        one device allocation per call.  Files outside the supported subset come back as ``None``
        (decode those with PIL)."""
        dev = torch.device(device).index if device is not None else None
        dev = torch.cuda.current_device() if dev is None else dev
        n = len(datas)
        if n == 0:
            return []
        ptrs, lens = (C.c_void_p * n)(), (C.c_size_t * n)()
        for i, d in enumerate(datas):
            if isinstance(d, torch.Tensor):
                if d.dtype != torch.uint8 or d.dim() != 1 or d.is_cuda or not d.is_contiguous():
                    raise ValueError('JPEG data tensors must be contiguous 1-D uint8 CPU tensors')
                ptrs[i], lens[i] = d.data_ptr(), d.numel()
            else:
                ptrs[i], lens[i] = C.cast(C.c_char_p(d), C.c_void_p), len(d)  # borrows d's buffer
        sizes, offsets, total = [], [], 0
        hs, ws, st = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
        self._lib.oake_jpeg_info_batch(n, ptrs, lens, hs, ws, st)
        for i in range(n):
            ok = st[i] == _lib.OAKE_OK
            sizes.append((hs[i], ws[i]) if ok else None)
            offsets.append(total)
            if ok:
                total += (hs[i] * ws[i] * 3 + 255) & ~255
        arena = torch.empty(max(total, 1), dtype=torch.uint8, device=torch.device('cuda', dev))
        outs = [arena[o:o + s[0] * s[1] * 3].view(s[0], s[1], 3) if s else None for s, o in zip(sizes, offsets)]
        base = arena.data_ptr()
        optr = (C.c_void_p * n)(*[base + o if s else None for s, o in zip(sizes, offsets)])
        caps = (C.c_size_t * n)(*[s[0] * s[1] * 3 if s else 0 for s in sizes])
        status = (C.c_int * n)()
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            rc = self._lib.oake_decode_jpeg_batch(h, n, ptrs, lens, optr, caps, None, None, status, threads,
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_decode_jpeg_batch')
        return [o if status[i] == _lib.OAKE_OK else None for i, o in enumerate(outs)]

    def crop_normalize(self, image_u8: torch.Tensor, boxes_xyxy, *,
                       out_dtype: torch.dtype = torch.float32,
                       out: torch.Tensor | None = None) -> torch.Tensor:
        """Exact-size (n x n) integer crops + ToTensor + Normalize (blocks of one pyramid level)."""
        from .preprocess import CLIP_MEAN, CLIP_STD
        image_u8, dev = self._image_args(image_u8)
        boxes = torch.as_tensor(boxes_xyxy, dtype=torch.int32).reshape(-1, 4).to(image_u8.device).contiguous()
        k, n = boxes.shape[0], self.input_resolution
        out = self._crop_out(out, k, n, out_dtype, image_u8.device)
        if k == 0:
            return out
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            mean, std = (C.c_float * 3)(*CLIP_MEAN), (C.c_float * 3)(*CLIP_STD)
            rc = self._lib.oake_crop_normalize(h, image_u8.data_ptr(), image_u8.shape[0], image_u8.shape[1],
                                               boxes.data_ptr(), k, n, mean, std, out.data_ptr(),
                                               _TORCH2OAKE[out_dtype],
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_crop_normalize')
        return out

    # -- profiler (bench.py) ---------------------------------------------------------------
    def profile(self, enable: bool | int) -> None:
        """``True`` / 1: stamp every launch; an int S > 1: every S-th launch only (``oake_profile_enable``)."""
        if self._handle is None:
            raise RuntimeError('run one forward before enabling the profiler')
        self._lib.oake_profile_reset(self._handle)
        self._lib.oake_profile_enable(self._handle, int(enable))

    def profile_read(self) -> list[dict]:
        n = C.c_int(0)
        buf = (_lib.ProfileEntry * 64)()
        _lib.check(self._lib, self._handle,
                   self._lib.oake_profile_read(self._handle, buf, 64, C.byref(n)), 'profile_read')
        return [dict(name=buf[i].name.decode(), total_ms=buf[i].total_ms, flops=buf[i].flops,
                     bytes=buf[i].bytes, launches=buf[i].launches, seen=buf[i].seen) for i in range(min(n.value, 64))]


class TextTransformer:
    """``clip.model.CLIP``'s text tower behind ``oake_encode_text`` (oadp/prompts/vild.py:62-66)."""

    NAMES = ('token_embedding.weight', 'positional_embedding', 'ln_final.weight', 'ln_final.bias',
             'text_projection')

    def __init__(self, state_dict: Mapping[str, torch.Tensor], *, compute_dtype=torch.float16,
                 max_batch: int = 256, **_ignored) -> None:
        self._sd = {k: v.detach().to('cpu', torch.float32).contiguous() for k, v in state_dict.items()
                    if k in self.NAMES or k.startswith('transformer.resblocks.')}
        sd = self._sd
        self.vocab, self.width = sd['token_embedding.weight'].shape
        self.context = sd['positional_embedding'].shape[0]
        self.layers = len({k.split('.')[2] for k in sd if k.startswith('transformer.resblocks.')})
        self.heads = self.width // 64
        self.mlp_dim = sd['transformer.resblocks.0.mlp.c_fc.weight'].shape[0]
        self.output_dim = sd['text_projection'].shape[1]
        self.compute_dtype = compute_dtype
        self.max_batch = max_batch
        self._lib = _lib.load()
        self._handle = None
        self._handle_dev = None

    def _ensure_handle(self, device_index: int):
        if self._handle is not None and self._handle_dev == device_index:
            return self._handle
        self.close()
        lib = self._lib
        cfg = _lib.OakeTextConfig()
        lib.oake_text_default_config(C.byref(cfg))
        cfg.context, cfg.vocab, cfg.width, cfg.layers = self.context, self.vocab, self.width, self.layers
        cfg.heads, cfg.mlp_dim, cfg.embed_dim = self.heads, self.mlp_dim, self.output_dim
        cfg.compute_dtype = _TORCH2OAKE[self.compute_dtype]
        cfg.max_batch = self.max_batch
        h = C.c_void_p()
        _lib.check(lib, None, lib.oake_text_create(C.byref(cfg), device_index, C.byref(h)), 'oake_text_create')
        try:
            for name, t in self._sd.items():
                _lib.check(lib, h, lib.oake_load_tensor(h, name.encode(), t.data_ptr(), t.numel()),
                           f'oake_load_tensor({name})')
            missing = lib.oake_missing_tensors(h)
            if missing:
                raise ValueError(f'state_dict lacks {missing} text-tower tensors')
        except Exception:
            lib.oake_destroy(h)
            raise
        self._handle, self._handle_dev = h, device_index
        return h

    def close(self) -> None:
        if self._handle is not None:
            self._lib.oake_destroy(self._handle)
            self._handle = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, tokens: torch.Tensor, *, normalize: bool = False,
                 out_dtype: torch.dtype | None = None) -> torch.Tensor:
        if not tokens.is_cuda:
            raise RuntimeError('oadp_amd.clip runs on the GPU only (no CPU fallback)')
        if tokens.dim() != 2 or not 1 <= tokens.shape[1] <= self.context:
            raise ValueError(f'expected [N, L <= {self.context}] token ids, got {tuple(tokens.shape)}')
        tokens = tokens.to(torch.int32).contiguous()
        out_dtype = out_dtype or self.compute_dtype
        dev = tokens.device.index if tokens.device.index is not None else torch.cuda.current_device()
        out32 = out_dtype != torch.float16
        out = torch.empty((tokens.shape[0], self.output_dim), device=tokens.device,
                          dtype=torch.float32 if out32 else torch.float16)
        with torch.cuda.device(dev):
            h = self._ensure_handle(dev)
            rc = self._lib.oake_encode_text(h, tokens.data_ptr(), tokens.shape[0], tokens.shape[1],
                                            out.data_ptr(), _lib.OAKE_F32 if out32 else _lib.OAKE_F16,
                                            int(normalize), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(self._lib, h, rc, 'oake_encode_text')
        return out.to(out_dtype)


class CLIP:
    """``clip.model.CLIP`` as the reference uses it: ``visual`` / ``encode_image`` for OAKE, and
    ``encode_text`` (oadp/prompts/vild.py) when the state dict carries the text tower."""

    def __init__(self, state_dict: Mapping[str, torch.Tensor], **kwargs) -> None:
        self.visual = VisionTransformer(state_dict, **kwargs) if 'visual.conv1.weight' in state_dict else None
        self.text = TextTransformer(state_dict, **kwargs) if 'token_embedding.weight' in state_dict else None

    def encode_text(self, text: torch.Tensor, **kwargs) -> torch.Tensor:
        if self.text is None:
            raise RuntimeError('this checkpoint has no text tower')
        return self.text(text, **kwargs)

    @property
    def dtype(self) -> torch.dtype:
        return (self.visual or self.text).compute_dtype

    def encode_image(self, image: torch.Tensor, **kwargs) -> torch.Tensor:
        # reference: self.visual(image.type(self.dtype)); the cast happens inside the im2col kernel
        return self.visual(image, **kwargs)

    def eval(self) -> 'CLIP':
        return self

    def requires_grad_(self, flag: bool = False) -> 'CLIP':
        return self


DEFAULT_CHECKPOINT = 'pretrained/clip/ViT-B-32.pt'  # reference README.md:129


def _read_checkpoint(path: str | os.PathLike) -> dict[str, torch.Tensor]:
    p = pathlib.Path(path)
    try:  # OpenAI checkpoints are TorchScript archives
        return dict(torch.jit.load(str(p), map_location='cpu').state_dict())
    except RuntimeError:
        obj = torch.load(str(p), map_location='cpu')
        return dict(obj.get('state_dict', obj))


def load(state_dict: Mapping[str, torch.Tensor] | str | os.PathLike, *, squash: bool = False,
         **kwargs) -> tuple[CLIP, Preprocess]:
    if not isinstance(state_dict, Mapping):
        state_dict = _read_checkpoint(state_dict)
    model = CLIP(state_dict, **kwargs)
    return model, Preprocess(model.visual.input_resolution if model.visual else 224, squash=squash)


def load_default(flag: bool = False, **kwargs) -> tuple[CLIP, Preprocess]:
    """``clip.load_default(flag)`` of the fork: ViT-B/32 + its transform.  ``flag`` selects the
    transform variant (see Preprocess; what True means is the ``fork.load_default_true`` setting,
    oadp_amd/clip/settings.py).  Weights: ``$OAKE_CLIP_CHECKPOINT`` or
    pretrained/clip/ViT-B-32.pt; with ``OAKE_SYNTHETIC_WEIGHTS=1`` (or DRY_RUN=True and no
    checkpoint on disk) the deterministic synthetic ViT-B/32 of oadp_amd.weights is used."""
    path = os.environ.get('OAKE_CLIP_CHECKPOINT', DEFAULT_CHECKPOINT)
    synthetic = os.environ.get('OAKE_SYNTHETIC_WEIGHTS', '') not in ('', '0', 'False')
    dry = os.environ.get('DRY_RUN', '') not in ('', '0', 'False')
    squash = bool(flag) and settings.load_default_true == 'squash'
    if not synthetic and os.path.exists(path):
        return load(path, squash=squash, **kwargs)
    if synthetic or dry:
        from ..weights import synthetic_state_dict
        return load(synthetic_state_dict(), squash=squash, **kwargs)
    raise FileNotFoundError(f'{path} not found (set OAKE_CLIP_CHECKPOINT, or '
                            'OAKE_SYNTHETIC_WEIGHTS=1 for random-init weights)')
