"""The C-ABI library loads on a CPU-only box and exports every symbol include/oake_hip.h declares
(no compute calls here — those are the -m gpu tests)."""
import ctypes as C
import pathlib
import re
import subprocess

import pytest

from oadp_amd import _lib

ROOT = pathlib.Path(__file__).resolve().parents[1]
HEADER = ROOT / 'include' / 'oake_hip.h'              # the reference-facing ABI
DEBUG_HEADER = ROOT / 'include' / 'oake_hip_debug.h'  # kernel-level test / measurement entry points


def _declared(header=HEADER):
    text = header.read_text()
    return sorted(set(re.findall(r'OAKE_API[^;]*?\b(oake_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 20
    out = subprocess.run(['nm', '-D', '--defined-only', str(_lib.LIB_PATH)], capture_output=True,
                         text=True, check=True).stdout
    exported = set(re.findall(r' T (oake_[a-z0-9_]+)', out))
    assert set(names) <= exported, set(names) - exported
    # and the Python binding covers exactly the header
    assert set(_lib.SIGNATURES) == set(names)
    for n in names:
        assert getattr(lib, n) is not None
    # the public header carries no lab bench: every oake_debug_* lives in oake_hip_debug.h
    assert not [n for n in names if n.startswith('oake_debug_')]
    debug = _declared(DEBUG_HEADER)
    assert debug and all(n.startswith('oake_debug_') for n in debug)
    assert set(debug) <= exported, set(debug) - exported
    assert set(_lib.DEBUG_SIGNATURES) == set(debug)
    assert exported == set(names) | set(debug), exported ^ (set(names) | set(debug))


def test_abi_version_and_default_config(lib):
    assert lib.oake_abi_version() == _lib.ABI_VERSION
    cfg = _lib.OakeConfig()
    lib.oake_default_config(C.byref(cfg))
    assert (cfg.image_size, cfg.patch_size, cfg.stride, cfg.padding) == (224, 32, 32, 0)
    assert (cfg.width, cfg.layers, cfg.heads, cfg.mlp_dim, cfg.embed_dim) == (768, 12, 12, 3072, 512)
    assert cfg.compute_dtype == _lib.OAKE_F16 and cfg.max_batch == 256
    assert cfg.residual_dtype == _lib.OAKE_F16 and cfg.pass_rows == 0
    assert C.sizeof(_lib.OakeConfig) == 52


def test_create_rejects_bad_config_without_gpu(lib):
    cfg = _lib.OakeConfig()
    lib.oake_default_config(C.byref(cfg))
    cfg.width = 100  # not heads * 64
    h = C.c_void_p()
    assert lib.oake_create(C.byref(cfg), 0, C.byref(h)) == 1  # OAKE_ERR_INVALID, before any HIP call
    assert b'width' in lib.oake_last_error(None)


def test_no_cpu_fallback_in_product():
    """The product path never imports the oracle and refuses CPU tensors."""
    import torch
    from oadp_amd import clip
    from oadp_amd.weights import synthetic_state_dict
    for p in (ROOT / 'oadp_amd').rglob('*.py'):
        assert 'oracle' not in p.read_text().replace('oracle/', ''), p
    model, _ = clip.load(synthetic_state_dict(width=128, layers=1, heads=2, mlp_dim=256, embed_dim=64))
    try:
        model.encode_image(torch.zeros(1, 3, 224, 224))
    except RuntimeError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('CPU tensor was accepted')


def test_cu_half_masks_are_complementary_and_default_exists():
    """half_masks() default scheme must exist; the two masks partition the CUs."""
    from oadp_amd import cumask
    for scheme in (None, 'halves', 'even_odd', 'group8', 'pairs', 'group4'):
        a, b = cumask.half_masks(256) if scheme is None else cumask.half_masks(256, scheme)
        assert len(a) == len(b) == 8
        for wa, wb in zip(a, b):
            assert wa & wb == 0 and (wa | wb) == 0xFFFFFFFF
        assert sum(bin(w).count('1') for w in a) == 128
    assert cumask.half_masks(256, 'pairs') == cumask.half_masks(256, 'group8')
    with pytest.raises(ValueError):
        cumask.half_masks(256, 'nonsense')


def test_lab_library_is_a_superset_build_and_the_product_is_smaller(lib):
    """liboake_hip_lab.so (same sources, -DOAKE_LAB=1) exports exactly the same symbols; the production library is
    the smaller one and says it is not the lab build; nothing in the product path names the lab library."""
    lab = _lib.load_lab()
    def exported(path):
        out = subprocess.run(['nm', '-D', '--defined-only', str(path)], capture_output=True, text=True, check=True).stdout
        return set(re.findall(r' T (oake_[a-z0-9_]+)', out))
    assert exported(_lib.LAB_PATH) == exported(_lib.LIB_PATH)
    assert lib.oake_debug_lab_build() == 0 and lab.oake_debug_lab_build() == 1
    assert _lib.LIB_PATH.stat().st_size < 0.6 * _lib.LAB_PATH.stat().st_size
    for p in (ROOT / 'oadp_amd').rglob('*.py'):
        if p.name not in ('_lib.py', 'build.py'):
            assert 'load_lab' not in p.read_text() and 'liboake_hip_lab' not in p.read_text(), p


def test_product_library_reads_no_environment_and_carries_no_shelved_kernel():
    """(a) the pass size decides output rounding, so nothing inside liboake_hip.so may take it (or
    anything else) from the process environment — the library imports no getenv; the Python host maps OAKE_PASS_ROWS /
    OAKE_PASS_CROPS onto oake_config (clip/model.py::_pass_config).  (b) the three-image fused kernel (csrc/qkv_attn.hip)
    lost its A/B and is linked into the lab build only."""
    und = subprocess.run(['nm', '-D', '--undefined-only', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in und, [ln for ln in und.splitlines() if 'getenv' in ln]
    syms = subprocess.run(['nm', '-C', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    assert 'qkv_attn_kernel' not in syms and 'attn_out_kernel' not in syms
    lab = subprocess.run(['nm', '-C', str(_lib.LAB_PATH)], capture_output=True, text=True, check=True).stdout
    assert 'qkv_attn_kernel' in lab and 'attn_out_kernel' in lab
    for src in (ROOT / 'oadp_amd' / 'csrc').glob('*'):
        if src.is_file():
            assert 'getenv' not in src.read_text(), src


def test_pass_config_maps_the_environment_onto_the_config():
    from oadp_amd.clip.model import _pass_config
    assert _pass_config(512, {}) == (512, 0)                           # the library's default: 25 600 rows
    assert _pass_config(512, {'OAKE_PASS_ROWS': '0'}) == (512, -1)     # no row cap
    assert _pass_config(512, {'OAKE_PASS_ROWS': '29800'}) == (512, 29800)
    assert _pass_config(512, {'OAKE_PASS_CROPS': '120'}) == (120, -1)  # the cap in crops, no row cap beside it
    assert _pass_config(64, {'OAKE_PASS_CROPS': '120'}) == (64, -1)


def test_pass_planner_host_arithmetic(lib):
    """How a call's crops are cut into encoder passes (csrc/api.hip plan_pass_size; no GPU involved): never above the cap,
    never more passes than two beyond the fewest that fit, one pass when everything fits, and the two cases its comment
    quotes on a 256-CU chip."""
    plan = lambda cap, n, tokens, per_tile, ncu=256: lib.oake_debug_plan_pass(cap, n, tokens, per_tile, 768, 3072, 12, ncu)
    assert plan(512, 256, 50, 4) == 256 and plan(512, 512, 50, 4) == 512 and plan(512, 1, 50, 4) == 1
    # blocks mode, 64 images of 640 x 480 = 1728 crops: three full passes and a short one fill whole rounds of every
    # kernel; four equal passes of 432 leave c_proj / out_proj on 80 % of the CUs
    assert plan(512, 1728, 50, 4) == 512
    # a remainder that would run on a sliver of the chip is not left alone: 513 crops go as two equal passes
    assert plan(512, 513, 50, 4) == 257
    # objects mode (197 tokens per crop, cap 129): 2400 proposal crops of 8 images
    per = plan(129, 2400, 197, 1)
    assert 100 <= per <= 129
    for cap, n, tokens, per_tile in [(512, 15680, 50, 4), (129, 601, 197, 1), (64, 1000, 50, 4), (7, 50, 197, 1), (1, 5, 50, 4)]:
        for ncu in (256, 128, 0):
            per = plan(cap, n, tokens, per_tile, ncu)
            k0 = -(-n // cap)
            assert 1 <= per <= cap and k0 <= -(-n // per) <= k0 + 2, (cap, n, tokens, per, ncu)
