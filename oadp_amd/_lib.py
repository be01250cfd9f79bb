"""ctypes binding of liboake_hip.so (include/oake_hip.h).  No CPU fallback: if the HIP library
is missing or fails to load, importing a model raises — the product path never routes around it."""
from __future__ import annotations

import ctypes as C
import os
import pathlib

OAKE_OK = 0
OAKE_ERR_INVALID, OAKE_ERR_HIP, OAKE_ERR_STATE, OAKE_ERR_UNKNOWN_TENSOR, OAKE_ERR_UNSUPPORTED = 1, 2, 3, 4, 5
OAKE_F32, OAKE_F16, OAKE_BF16, OAKE_U8 = 0, 1, 2, 3
OAKE_LAYOUT_PADDED = 0x100
OAKE_OPT_CLS_LAST, OAKE_OPT_GEMM_VARIANT, OAKE_OPT_GEMM_PANEL, OAKE_OPT_ATTENTION_VARIANT = 1, 2, 3, 4
OAKE_OPT_PATCH_DIRECT = 5
OAKE_OPT_CU_COUNT = 6
OAKE_OPT_FUSE_ATTN_OUT = 7
OAKE_OPT_PASS_CROPS = 8
OAKE_OPT_FUSE_QKV_ATTN = 9
OAKE_OPT_QKV_WALK = 10
ABI_VERSION = 4

# OAKE_LIB: kernel-experiment builds (tools/); the product always loads the in-tree library
LIB_PATH = pathlib.Path(os.environ.get('OAKE_LIB') or pathlib.Path(__file__).resolve().parent / 'liboake_hip.so')


class OakeConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'image_size', 'patch_size', 'stride', 'padding', 'width', 'layers', 'heads', 'mlp_dim',
        'embed_dim', 'compute_dtype', 'max_batch', 'residual_dtype', 'pass_rows')]


class ProfileEntry(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('total_ms', C.c_double), ('flops', C.c_double),
                ('bytes', C.c_double), ('launches', C.c_int64), ('seen', C.c_int64)]


# name -> (restype, argtypes); every symbol include/oake_hip.h declares
_VP, _I = C.c_void_p, C.c_int
SIGNATURES = {
    'oake_abi_version': (C.c_uint32, []),
    'oake_default_config': (None, [C.POINTER(OakeConfig)]),
    'oake_create': (_I, [C.POINTER(OakeConfig), _I, C.POINTER(_VP)]),
    'oake_destroy': (None, [_VP]),
    'oake_last_error': (C.c_char_p, [_VP]),
    'oake_grid': (_I, [_VP]),
    'oake_tokens': (_I, [_VP]),
    'oake_padded_layout': (_I, [_VP, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    'oake_load_tensor': (_I, [_VP, C.c_char_p, _VP, C.c_size_t]),
    'oake_missing_tensors': (_I, [_VP]),
    'oake_encode_image': (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP]),
    'oake_encode_objects': (_I, [_VP, _VP, _I, _VP, _I, _I, _VP, _I, _I, _VP]),
    'oake_crop_normalize': (_I, [_VP, _VP, _I, _I, _VP, _I, _I, C.POINTER(C.c_float),
                                 C.POINTER(C.c_float), _VP, _I, _VP]),
    'oake_crop_resize_normalize': (_I, [_VP, _VP, _I, _I, C.POINTER(C.c_float), _I, _I, _I,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float), _VP, _I, _VP]),
    'oake_crop_resize_normalize_batch': (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _I, _I, C.POINTER(C.c_float),
                                              C.POINTER(C.c_float), _VP, _I, _VP]),
    'oake_blocks_batch': (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, C.c_double, C.POINTER(C.c_float),
                               C.POINTER(C.c_float), _VP, _I, _VP, _VP]),
    'oake_blocks_count': (_I, [_I, _I, _I, _I, C.c_double]),
    'oake_jpeg_info_batch': (_I, [_I, _VP, _VP, _VP, _VP, _VP]),
    'oake_resize_u8': (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP]),
    'oake_text_default_config': (None, [_VP]),
    'oake_text_create': (_I, [_VP, _I, C.POINTER(C.c_void_p)]),
    'oake_encode_text': (_I, [_VP, _VP, _I, _I, _VP, _I, _I, _VP]),
    'oake_jpeg_info': (_I, [_VP, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'oake_decode_jpeg_batch': (_I, [_VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    'oake_jpeg_entropy_decode': (_I, [_VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(C.c_size_t)]),
    'oake_jpeg_reconstruct': (_I, [_VP, _VP, C.c_size_t, _VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _VP]),
    'oake_decode_jpeg': (_I, [_VP, _VP, C.c_size_t, _VP, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), _VP]),
    'oake_profile_enable': (_I, [_VP, _I]),
    'oake_profile_read': (_I, [_VP, C.POINTER(ProfileEntry), _I, C.POINTER(_I)]),
    'oake_profile_reset': (_I, [_VP]),
    'oake_set_option': (_I, [_VP, _I, _I]),
    'oake_get_option': (_I, [_VP, _I, C.POINTER(_I)]),
}

# include/oake_hip_debug.h: kernel-level test / measurement entry points (tests/, tools/, bench.py's power
# probe) — exported by the same library, not part of the reference-facing ABI above
DEBUG_SIGNATURES = {
    'oake_debug_gemm': (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    'oake_debug_gemm16': (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    'oake_debug_ln_gemm16': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _I, _VP]),
    'oake_debug_layernorm': (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _VP]),
    'oake_debug_attention': (_I, [_VP, _VP, _I, _I, _I, _I, _VP]),
    'oake_debug_ln_qkv_attn_obj': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I, _I, _I, _I, _VP, _I, _VP]),
    'oake_debug_ln_qkv_attn': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _I, _VP]),
    'oake_debug_ln_qkv_attn_quad': (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _I, _VP]),
    'oake_debug_attention_objects': (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _I, _I, _I, _VP]),
    'oake_debug_attn_out': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    'oake_debug_attn_out_trace': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _I, _VP]),
    'oake_debug_tr_read': (_I, [_VP, _VP, _VP]),
    'oake_debug_cu_census': (_I, [_VP, _I, _I, _VP]),
    'oake_debug_mfma_probe': (_I, [_VP, _VP, C.c_int, C.POINTER(C.c_double), _VP]),
    'oake_debug_mfma_probe_order': (_I, [_VP, _VP, C.c_int, C.c_int, C.POINTER(C.c_double), _VP]),
    'oake_debug_mfma_probe_32x32': (_I, [_VP, _VP, C.c_int, C.POINTER(C.c_double), _VP]),
    'oake_debug_set_attention_variant': (_I, [_I]),
    'oake_debug_set_gemm_variant': (_I, [_I]),
    'oake_debug_lab_build': (_I, []),
    'oake_debug_plan_pass': (_I, [_I] * 8),
    'oake_debug_gemm_resid16': (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    'oake_debug_set_gemm_panel': (_I, [_I]),
    'oake_debug_set_qkv_walk': (_I, [_I]),
    'oake_debug_set_gemm_trace': (_I, [_VP]),
    'oake_debug_read_weight16': (_I, [_VP, C.c_char_p, _VP, C.c_size_t]),
}

_lib = None
_lab = None
# OAKE_LAB_LIB: a lab-flavoured experiment build (tools/ only), as OAKE_LIB for the product
LAB_PATH = pathlib.Path(os.environ.get('OAKE_LAB_LIB') or pathlib.Path(__file__).resolve().parent / 'liboake_hip_lab.so')


class OakeTextConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('context', 'vocab', 'width', 'layers', 'heads', 'mlp_dim',
                                         'embed_dim', 'compute_dtype', 'max_batch', 'r0', 'r1', 'r2')]


def load() -> C.CDLL:
    """Load liboake_hip.so.  ``import torch`` first so the HIP runtime torch ships (same SONAME,
    libamdhip64.so.7) is the one the library binds to — one runtime, shared streams/pointers."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads torch's libamdhip64 before ours is resolved)
    if not LIB_PATH.exists():
        raise ImportError(
            f'{LIB_PATH} not found: build it with `python -m oadp_amd.build` '
            '(or __graft_entry__.build()); there is no CPU fallback')
    lib = _bind(C.CDLL(str(LIB_PATH), mode=C.RTLD_LOCAL), tolerate_missing_debug=bool(os.environ.get('OAKE_LIB')))
    _lib = lib
    return lib


def _bind(lib: C.CDLL, tolerate_missing_debug: bool = False) -> C.CDLL:
    for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
        if name in DEBUG_SIGNATURES and tolerate_missing_debug and not hasattr(lib, name):
            continue  # an older experiment build (tools/ab_env.py) may lack a newer debug entry point
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.oake_abi_version() != ABI_VERSION:
        raise ImportError(f'{lib._name}: ABI {lib.oake_abi_version()} != {ABI_VERSION}')
    return lib


def load_lab() -> C.CDLL:
    """liboake_hip_lab.so: the production kernels plus the tile configurations / kernel forms / measurement
    epilogues that lost their A/B (built with -DOAKE_LAB=1 by oadp_amd.build).  For tools/ and the variant tests
    only — nothing in the product path loads it (pass ``lib=load_lab()`` to ``clip.load`` to drive a model on it)."""
    global _lab
    if _lab is None:
        import torch  # noqa: F401
        if not LAB_PATH.exists():
            raise ImportError(f'{LAB_PATH} not found: build it with `python -m oadp_amd.build`')
        _lab = _bind(C.CDLL(str(LAB_PATH), mode=C.RTLD_LOCAL))
    return _lab


class OakeError(RuntimeError):
    pass


def check(lib: C.CDLL, handle, rc: int, what: str) -> None:
    if rc != OAKE_OK:
        msg = lib.oake_last_error(handle)
        raise OakeError(f'{what} failed (status {rc}): {msg.decode() if msg else "?"}')
