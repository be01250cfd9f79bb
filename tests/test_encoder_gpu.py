"""End-to-end parity of the HIP encoder (through oadp_amd.clip -> C ABI) against the fp32 CPU
oracle on the same seeded weights and inputs.  Tolerance = BASELINE.json north_star: fp16
rtol 1e-3 / atol 1e-3 on L2-normalised features and cosine >= 0.999."""
import os

import pytest
import torch

from oadp_amd import clip
from oadp_amd.weights import synthetic_images, synthetic_state_dict
from oracle.vit_ref import ViTConfig, encode_image_ref, encode_objects_ref, l2_normalize

pytestmark = pytest.mark.gpu

TINY = dict(width=128, layers=2, heads=2, mlp_dim=512, embed_dim=64)


def _check(out, ref, rtol, atol, cos_min=0.999):
    out = out.float().cpu()
    cos = torch.nn.functional.cosine_similarity(out, ref, dim=1)
    err = (out - ref).abs().max().item()
    print(f'max|err|={err:.3e} min cos={cos.min().item():.6f}')
    assert cos.min().item() >= cos_min
    torch.testing.assert_close(out, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize('resid32', [False, True])
@pytest.mark.parametrize('dtype,rtol,atol', [(torch.float16, 1e-3, 1e-3), (torch.bfloat16, 2e-2, 8e-3)])
@pytest.mark.parametrize('n', [1, 5, 27])
def test_encode_image_tiny(cuda, dtype, rtol, atol, n, resid32):
    sd = synthetic_state_dict(**TINY)
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=16,
                         residual_dtype=torch.float32 if resid32 else None)
    x = synthetic_images(n, seed=n)
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(**TINY), x))
    out = model.encode_image(x.to(cuda), normalize=True, out_dtype=torch.float32)
    _check(out, ref, rtol, atol)
    # un-normalised surface (what the reference's encode_image returns), f16 like model.dtype
    raw = model.encode_image(x.to(cuda))
    assert raw.dtype == dtype and raw.shape == (n, TINY['embed_dim'])
    _check(torch.nn.functional.normalize(raw.float()), ref, 3 * rtol, 3 * atol)


@pytest.mark.parametrize('resid32', [False, True])
def test_encode_image_vit_b32(cuda, resid32):
    """Full ViT-B/32 (SURVEY.md §3.4 constants), 12 layers, 87.8 M parameters; fp16 residual stream
    (default, like the reference's fp16 GPU model) and fp32 residual stream."""
    sd = synthetic_state_dict()
    model, _ = clip.load(sd, max_batch=8, residual_dtype=torch.float32 if resid32 else None)
    x = synthetic_images(11, seed=3)  # 11 > max_batch: exercises the multi-pass path
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x))
    out = model.encode_image(x.to(cuda), normalize=True, out_dtype=torch.float16)
    assert out.dtype == torch.float16 and out.shape == (11, 512)
    _check(out, ref, 1e-3, 1e-3)
    # half-precision input (the reference casts images to model.dtype before conv1)
    out16 = model.encode_image(x.half().to(cuda), normalize=True, out_dtype=torch.float32)
    ref16 = l2_normalize(encode_image_ref(sd, ViTConfig(), x.half().float()))
    _check(out16, ref16, 1e-3, 1e-3)


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 2e-2)])
def test_encode_image_vit_b32_production_kernels(cuda, dtype, tol):
    """48 crops x 50 tokens = 2400 rows: large enough that every main-stream GEMM runs the
    persistent kernel with the LayerNorm folded in and the row statistics handed from the residual
    epilogue to the next GEMM (csrc/gemm.hip) — the path bench.py measures; 45 crops leave a ragged
    last tile (2250 = 14 * 160 + 10)."""
    sd = synthetic_state_dict()
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=48)
    for n in (48, 45):
        x = synthetic_images(n, seed=100 + n)
        ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x))
        out = model.encode_image(x.to(cuda), normalize=True, out_dtype=torch.float32)
        _check(out, ref, tol, tol)


def test_encode_image_full_size_properties(cuda):
    """BASELINE.json configs[1] at its full size (ViT-B/32, 256 crops): size-independent properties.
    Splitting or permuting the batch must not change any image's feature bit-wise (rows are
    independent through every kernel; all of these shapes run the same persistent kernels), the
    features are unit vectors, and a subset agrees with the oracle."""
    sd = synthetic_state_dict()
    model, _ = clip.load(sd, max_batch=256)
    x = synthetic_images(256, seed=77).to(cuda)
    full = model.encode_image(x, normalize=True, out_dtype=torch.float16)
    assert full.shape == (256, 512) and torch.isfinite(full.float()).all()
    assert (full.float().norm(dim=1) - 1).abs().max().item() < 2e-3
    halves = torch.cat([model.encode_image(x[:128], normalize=True, out_dtype=torch.float16),
                        model.encode_image(x[128:], normalize=True, out_dtype=torch.float16)])
    assert torch.equal(full, halves)
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(5)).to(cuda)
    assert torch.equal(model.encode_image(x[perm], normalize=True, out_dtype=torch.float16), full[perm])
    idx = [0, 63, 130, 255]
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x[idx].cpu()))
    _check(full[idx].float(), ref, 1e-3, 1e-3)


def test_encode_image_batch_invariance(cuda):
    """The per-image .pth contract: an image's feature must not depend on its batch."""
    sd = synthetic_state_dict(**TINY)
    model, _ = clip.load(sd, max_batch=64)
    x = synthetic_images(33, seed=9).to(cuda)
    full = model.encode_image(x, normalize=True, out_dtype=torch.float32)
    one = torch.cat([model.encode_image(x[i:i + 1], normalize=True, out_dtype=torch.float32)
                     for i in (0, 17, 32)])
    assert torch.equal(full[[0, 17, 32]], one)


def _objects_model(sd, arch, dtype=torch.float16, max_batch=8, resid32=False):
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=max_batch,
                         residual_dtype=torch.float32 if resid32 else None)
    v = model.visual
    # the reference's surgery, oadp/oake/objects.py:292-301
    v.positional_embedding = v.interpolate_positional_embedding((v.grid * 2,) * 2)
    v.grid *= 2
    v.conv1.stride = tuple(s // 2 for s in v.conv1.stride)
    v.conv1.padding = ((v.patch_size - 1) // 2,) * 2
    v.object_stream = True
    sd2 = dict(sd)
    sd2['visual.positional_embedding'] = v.positional_embedding
    cfg = ViTConfig(**arch, stride=v.conv1.stride[0], padding=v.conv1.padding[0])
    return model, sd2, cfg


@pytest.mark.parametrize('n', [1, 3, 11])
def test_encode_objects_tiny(cuda, n):
    sd = synthetic_state_dict(**TINY)
    model, sd2, cfg = _objects_model(sd, TINY)
    assert cfg.grid == 14 and cfg.tokens == 197
    x = synthetic_images(n, seed=20 + n)
    g = torch.Generator().manual_seed(n)
    masks = (torch.rand(n, 1, 14, 14, generator=g) > 0.4).float()
    masks[0] = 0  # an all-foreground crop
    ref = l2_normalize(encode_objects_ref(sd2, cfg, x, masks))
    out = model.visual(x.to(cuda), masks.to(cuda), normalize=True, out_dtype=torch.float32)
    _check(out, ref, 1e-3, 1e-3)


@pytest.mark.parametrize('resid32', [False, True])
def test_encode_objects_vit_b32(cuda, resid32):
    sd = synthetic_state_dict()
    model, sd2, cfg = _objects_model(sd, {}, max_batch=4, resid32=resid32)
    x = synthetic_images(5, seed=77)
    g = torch.Generator().manual_seed(5)
    masks = (torch.rand(5, 1, 14, 14, generator=g) > 0.5).float()
    ref = l2_normalize(encode_objects_ref(sd2, cfg, x, masks))
    out = model.visual(x.to(cuda), masks.half().to(cuda), normalize=True, out_dtype=torch.float16)
    _check(out, ref, 1e-3, 1e-3)


def test_encode_objects_vit_b32_production_kernels(cuda):
    """Objects mode where the work is big enough for the production path: 8 crops x 197 tokens = 1576
    rows (+8 object-token rows in the same matrices) run the persistent LN-folded GEMMs with the row
    statistics handed over, the cooperative attention kernel carries the object tokens on its idle
    wave; the second pass (4 crops) takes the small-problem kernels.  Both against the oracle."""
    sd = synthetic_state_dict()
    model, sd2, cfg = _objects_model(sd, {}, max_batch=8)
    x = synthetic_images(12, seed=31)
    g = torch.Generator().manual_seed(12)
    masks = (torch.rand(12, 1, 14, 14, generator=g) > 0.5).float()
    masks[3] = 0
    masks[7] = 1
    ref = l2_normalize(encode_objects_ref(sd2, cfg, x, masks))
    out = model.visual(x.to(cuda), masks.to(cuda), normalize=True, out_dtype=torch.float32)
    _check(out, ref, 1e-3, 1e-3)


def test_encode_objects_full_size_properties(cuda):
    """BASELINE.json configs[3] at its size (600 crops = 2 images x 300 proposals, mini-batch 512):
    splitting or permuting the crops must not change any crop's feature bit-wise."""
    sd = synthetic_state_dict()
    model, _, _ = _objects_model(sd, {}, max_batch=512)
    x = synthetic_images(600, seed=8).half().to(cuda)
    g = torch.Generator().manual_seed(600)
    masks = (torch.rand(600, 1, 14, 14, generator=g) > 0.5).half().to(cuda)
    full = model.visual(x, masks, normalize=True, out_dtype=torch.float16)
    assert full.shape == (600, 512) and torch.isfinite(full.float()).all()
    assert (full.float().norm(dim=1) - 1).abs().max().item() < 2e-3
    halves = torch.cat([model.visual(x[:300], masks[:300], normalize=True, out_dtype=torch.float16),
                        model.visual(x[300:], masks[300:], normalize=True, out_dtype=torch.float16)])
    assert torch.equal(full, halves)
    perm = torch.randperm(600, generator=torch.Generator().manual_seed(6)).to(cuda)
    assert torch.equal(model.visual(x[perm], masks[perm], normalize=True, out_dtype=torch.float16), full[perm])


def test_errors_are_loud(cuda):
    sd = synthetic_state_dict(**TINY)
    model, _ = clip.load(sd)
    with pytest.raises(RuntimeError):
        model.encode_image(synthetic_images(1))  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        model.encode_image(torch.zeros(1, 3, 100, 100, device=cuda))
    with pytest.raises(ValueError):
        model.visual(torch.zeros(1, 3, 224, 224, device=cuda), torch.zeros(1, 1, 14, 14, device=cuda))
    assert model.encode_image(torch.zeros(0, 3, 224, 224, device=cuda)).shape == (0, 64)


@pytest.mark.parametrize('resid32', [False, True])
@pytest.mark.parametrize('cfg,n', [(TINY, 7), (None, 48), (dict(TINY, layers=1), 3)])
def test_last_block_for_cls_rows_only_is_exact_elimination(cuda, cfg, n, resid32):
    """encode_image runs the last block's query / attention output / out_proj / MLP for the CLS rows only
    (ln_post reads nothing else).  Against the all-rows execution the reference performs: the same
    embeddings up to the rounding of different tile shapes, and both within tolerance of the oracle."""
    sd = synthetic_state_dict(**cfg) if cfg else synthetic_state_dict()
    model, _ = clip.load(sd, max_batch=64, residual_dtype=torch.float32 if resid32 else None)
    x = synthetic_images(n, seed=3).to(cuda)
    # a per-handle switch (oake_set_option): a second model in the same process keeps its default
    other, _ = clip.load(sd, max_batch=64, residual_dtype=torch.float32 if resid32 else None)
    model.visual.set_option('cls_last', 0)
    full = model.encode_image(x, normalize=True, out_dtype=torch.float32)
    cls = other.encode_image(x, normalize=True, out_dtype=torch.float32)
    model.visual.set_option('cls_last', 1)
    assert torch.equal(model.encode_image(x, normalize=True, out_dtype=torch.float32), cls)
    cos = torch.nn.functional.cosine_similarity(full, cls, dim=1)
    print(f'max|cls - full|={(full - cls).abs().max().item():.3e} min cos={cos.min().item():.7f}')
    assert cos.min().item() > 0.99999
    torch.testing.assert_close(cls, full, rtol=1e-3, atol=6e-4)
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(**cfg) if cfg else ViTConfig(), x.cpu()))
    _check(cls, ref, 1e-3, 1e-3)
    _check(full, ref, 1e-3, 1e-3)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_conv1_patch_gather_equals_im2col(cuda, dtype):
    """Input already in the compute type (what the reference hands to conv1 after `.type(model.dtype)`, and what
    the device preprocessing writes): the conv1 GEMM gathers its patch rows straight from the NCHW batch.
    Same operand bits, same kernel, same tile order as the im2col route => bit-identical features; and
    within tolerance of the oracle.  45 crops leave a ragged last tile, 300 crops run three passes."""
    sd = synthetic_state_dict()
    direct, _ = clip.load(sd, compute_dtype=dtype, max_batch=128)
    via_im2col, _ = clip.load(sd, compute_dtype=dtype, max_batch=128)
    via_im2col.visual.set_option('patch_direct', 0)
    for n in (128, 45, 300):
        x = synthetic_images(n if n <= 128 else 100, seed=7 + n).to(dtype)
        if n > 128:
            x = x.repeat(3, 1, 1, 1)
        xd = x.to(cuda)
        a = direct.encode_image(xd, normalize=True, out_dtype=torch.float32)
        b = via_im2col.encode_image(xd, normalize=True, out_dtype=torch.float32)
        assert torch.equal(a, b), n
        # a view with a 16-byte-misaligned base falls back to im2col and still agrees
        if n == 45:
            ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x[:6].float()))
            _check(a[:6], ref, *((1e-3, 1e-3) if dtype == torch.float16 else (2e-2, 8e-3)))
    prof = direct.visual
    prof.profile(True)
    direct.encode_image(synthetic_images(64, seed=1).to(dtype).to(cuda))
    names = {p_['name'] for p_ in prof.profile_read()}
    prof.profile(False)
    assert 'im2col' not in names and 'gemm_conv1' in names


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_conv1_fp32_input_gathered_without_im2col(cuda, dtype):
    """FP32 input — what the reference hands over (oadp/oake/globals.py:54-57: the transform's float32 crop,
    `.cuda()`), and the headline bench's input: the conv1 GEMM's DMA waves fetch the patch rows with ordinary
    loads, round them to the compute type and write the LDS image themselves.  Same rounding (RNE) and same
    operand bits as the im2col route => bit-identical features; no `im2col` launch in the profile.  128 crops =
    full tiles, 45 = a ragged last tile (row clamp), 300 = three passes; one crop = the small-problem fallback."""
    sd = synthetic_state_dict()
    from oadp_amd import _lib
    # (opt-in, measured slower than im2col + GEMM with two lanes: the lab build carries it, the product refuses it)
    direct, _ = clip.load(sd, compute_dtype=dtype, max_batch=128, lib=_lib.load_lab())
    direct.visual.set_option('patch_direct', 2)
    prod, _ = clip.load(sd, compute_dtype=dtype, max_batch=4)
    prod.encode_image(synthetic_images(1, seed=1).to(cuda))
    with pytest.raises(_lib.OakeError):
        prod.visual.set_option('patch_direct', 2)
    via_im2col, _ = clip.load(sd, compute_dtype=dtype, max_batch=128)
    via_im2col.visual.set_option('patch_direct', 0)
    for n in (128, 45, 300, 1):
        x = synthetic_images(n if n <= 128 else 100, seed=17 + n)
        if n > 128:
            x = x.repeat(3, 1, 1, 1)
        x = x * 3.0 + 0.25  # values that do not sit on the fp16 grid
        xd = x.to(cuda)
        assert xd.dtype == torch.float32
        a = direct.encode_image(xd, normalize=True, out_dtype=torch.float32)
        b = via_im2col.encode_image(xd, normalize=True, out_dtype=torch.float32)
        assert torch.equal(a, b), n
        if n == 45:
            ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x[:6]))
            _check(a[:6], ref, *((1e-3, 1e-3) if dtype == torch.float16 else (2e-2, 8e-3)))
    # a view whose base is not 16-byte aligned falls back to im2col and still agrees
    buf = torch.zeros(64 * 3 * 224 * 224 + 1, device=cuda)
    xs = synthetic_images(64, seed=3).to(cuda)
    mis = buf[1:].view(64, 3, 224, 224)
    mis.copy_(xs)
    assert mis.data_ptr() % 16 != 0
    assert torch.equal(direct.encode_image(mis, normalize=True), direct.encode_image(xs, normalize=True))
    prof = direct.visual
    prof.profile(True)
    direct.encode_image(synthetic_images(64, seed=1).to(cuda))
    names = {p_['name'] for p_ in prof.profile_read()}
    prof.profile(False)
    assert 'im2col' not in names and 'gemm_conv1' in names


def test_pass_cap_and_equal_passes_are_invisible(cuda):
    """oake_create caps an encoder pass at ~25.6 k token rows (128 crops at 197 tokens) and a call is cut into equal
    passes (300 crops: 3 x 100, not 128 + 128 + 44): a crop's embedding must not depend on either beyond the rounding
    of the last layer's object-token GEMMs (whose tile shape follows the rows of the pass) — objects mode, ViT-B/32
    widths, against the same crops sent down in calls of 37 (one pass each), against a handle whose cap is switched
    off (one pass of 300), and against the oracle."""
    arch = dict(width=768, layers=2, heads=12, mlp_dim=3072, embed_dim=512)
    sd = synthetic_state_dict(**arch)
    model, sd2, cfg = _objects_model(sd, arch, max_batch=512)
    x = synthetic_images(60, seed=4).half().repeat(5, 1, 1, 1).to(cuda)  # 300 crops
    g = torch.Generator().manual_seed(3)
    masks = (torch.rand(300, 1, 14, 14, generator=g) < 0.3).half().to(cuda)
    a = model.visual(x, masks, normalize=True, out_dtype=torch.float32)
    b = torch.cat([model.visual(x[i:i + 37], masks[i:i + 37], normalize=True, out_dtype=torch.float32)
                   for i in range(0, 300, 37)])
    os.environ['OAKE_PASS_ROWS'] = '0'
    try:
        uncapped, _, _ = _objects_model(sd, arch, max_batch=512)
        c = uncapped.visual(x, masks, normalize=True, out_dtype=torch.float32)  # one pass of 300
    finally:
        del os.environ['OAKE_PASS_ROWS']
    print(f'passes of 100 vs 37: max|d|={(a - b).abs().max().item():.2e}; vs one pass of 300: {(a - c).abs().max().item():.2e}')
    torch.testing.assert_close(a, b, rtol=0, atol=3e-4)
    torch.testing.assert_close(a, c, rtol=0, atol=3e-4)
    ref = l2_normalize(encode_objects_ref(sd2, cfg, x[:4].float().cpu(), masks[:4].float().cpu()))
    _check(a[:4], ref, 1e-3, 1e-3)


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 2e-2)])
def test_encode_image_fused_qkv_attention(cuda, lab, dtype, tol):
    """ln_1 + in_proj + attention as one kernel per layer (csrc/qkv_attn.hip, OAKE_OPT_FUSE_QKV_ATTN) on the full
    ViT-B/32: against the oracle, against the two-launch form (the same 16-bit q / k / v values: the features agree to
    the rounding of different summation orders), that it really is the path taken (profile slot names), and that an
    image's result does not depend on its position in its tile or in the batch.  Value 1 takes the four-images-per-tile
    form at L = 50 (csrc/qkv_attn_obj.hip's QUAD form), value 2 the three-image one: switching re-permutes the folded
    in-projection (the forms read it in different column orders)."""
    sd = synthetic_state_dict()
    # (the lab build: value 2's three-image kernel is not in the production library, which must refuse the value)
    prod, _ = clip.load(sd, compute_dtype=dtype, max_batch=4)
    prod.encode_image(synthetic_images(2, seed=1).to(cuda))
    with pytest.raises(Exception, match='lab'):
        prod.visual.set_option('fuse_qkv_attn', 2)
    prod.visual.close()
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=48, lib=lab)
    x = synthetic_images(46, seed=146)  # 46 = 15 groups of three + one: a ragged last tile
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x))
    v = model.visual
    xg = x.to(cuda)
    model.encode_image(xg[:2])  # creates the handle
    v.set_option('fuse_qkv_attn', 1)
    v.profile(True)
    fused = model.encode_image(xg, normalize=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    names = {p['name']: p['launches'] for p in v.profile_read() if p['launches'] > 0}
    v.profile(False)
    assert names.get('qkv_attn') == 11 and 'attention' not in names and 'gemm_qkv' not in names, names
    _check(fused, ref, tol, tol)
    v.set_option('fuse_qkv_attn', 0)
    v.profile(True)
    plain = model.encode_image(xg, normalize=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    names = {p['name']: p['launches'] for p in v.profile_read() if p['launches'] > 0}
    v.profile(False)
    assert 'qkv_attn' not in names and names.get('attention') == 11 and names.get('gemm_qkv') == 11, names
    _check(plain, ref, tol, tol)
    assert (fused - plain).abs().max().item() <= tol
    v.set_option('fuse_qkv_attn', 2)
    v.profile(True)
    triple = model.encode_image(xg, normalize=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    names = {p['name']: p['launches'] for p in v.profile_read() if p['launches'] > 0}
    v.profile(False)
    assert names.get('qkv_attn') == 11 and 'attention' not in names and 'gemm_qkv' not in names, names
    _check(triple, ref, tol, tol)
    assert (fused - triple).abs().max().item() <= tol
    v.set_option('fuse_qkv_attn', 1)
    again = model.encode_image(xg.flip(0), normalize=True, out_dtype=torch.float32).flip(0)
    assert (again - fused).abs().max().item() <= 3e-4  # (the last layer's small GEMMs pick tiles by row position)
    with pytest.raises(Exception):
        v.set_option('fuse_qkv_attn', 3)


def test_pass_limit_is_a_memory_bound(cuda):
    """The reference's `mini_batch_size` [REF oadp/oake/objects.py:321-331] bounds the crops of one encoder pass.  Here:
    `visual.pass_limit` -> OAKE_OPT_PASS_CROPS lowers the library's cap (never raises it above what the handle's
    workspace was created for), existing and later handles alike; visible in the number of launches; the features move
    by the last layer's GEMM rounding only (advisor r04)."""
    arch = dict(width=768, layers=2, heads=12, mlp_dim=3072, embed_dim=512)
    sd = synthetic_state_dict(**arch)
    model, _ = clip.load(sd, max_batch=48)
    v = model.visual
    x = synthetic_images(45, seed=8).to(cuda)
    assert v.get_option('pass_crops') is None  # no handle yet
    a = model.encode_image(x, normalize=True, out_dtype=torch.float32)
    assert v.get_option('pass_crops') == 48

    def qkv_launches():
        v.profile(True)
        out = model.encode_image(x, normalize=True, out_dtype=torch.float32)
        torch.cuda.synchronize()
        d = {p['name']: p['launches'] for p in v.profile_read()}
        n = d.get('qkv_attn', 0) + d.get('gemm_qkv', 0)  # one per full layer and pass (fused with the attention where M > 1024)
        v.profile(False)
        return out, n

    _, one = qkv_launches()
    assert one == 1  # two layers, the last one runs for the CLS rows: one in-projection of all rows per pass
    v.pass_limit = 16
    assert v.get_option('pass_crops') == 16
    b, three = qkv_launches()  # 45 crops under a cap of 16: 3 passes of 15
    assert three == 3 * one and one >= 1
    torch.testing.assert_close(a, b, rtol=0, atol=3e-4)
    v.pass_limit = 1000  # a bound: not above the 48 the workspace was sized for
    assert v.get_option('pass_crops') == 48
    v.pass_limit = 16
    v.lane = 1  # a handle created later gets the same bound
    try:
        model.encode_image(x[:4])
        assert v.get_option('pass_crops') == 16
    finally:
        v.lane = 0
    with pytest.raises(Exception):
        v.set_option('pass_crops', 0)


def test_batches_beyond_1024_crops_per_pass(cuda, monkeypatch):
    """max_batch > 1024: the CLS rows of the last block (M = crops per pass) are then large enough for the
    persistent GEMM, which takes its LayerNorm statistics as per-row sums — decided per GEMM from the shape
    it is actually called with, not once per pass.  Same features as 256-crop passes."""
    arch = dict(width=768, layers=2, heads=12, mlp_dim=3072, embed_dim=512)  # ViT-B/32 widths, 2 layers
    sd = synthetic_state_dict(**arch)
    x = synthetic_images(100, seed=9).half().repeat(11, 1, 1, 1).to(cuda)  # 1100 crops, 100 distinct
    monkeypatch.setenv('OAKE_PASS_ROWS', '0')  # (no cap on the rows of a pass: this test is about M > 1024 CLS rows)
    big, _ = clip.load(sd, max_batch=1100)
    small, _ = clip.load(sd, max_batch=256)
    a = big.encode_image(x, normalize=True, out_dtype=torch.float32)
    b = small.encode_image(x, normalize=True, out_dtype=torch.float32)
    cos = torch.nn.functional.cosine_similarity(a, b, dim=1)
    print(f'max|a-b|={(a - b).abs().max().item():.3e} min cos={cos.min().item():.7f}')
    assert cos.min().item() > 0.99999
    torch.testing.assert_close(a, b, rtol=1e-3, atol=6e-4)
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(**arch), x[:6].float().cpu()))
    _check(a[:6], ref, 1e-3, 1e-3)


def test_lanes_are_independent_handles_on_their_own_streams(cuda):
    """visual.lane selects one of several native handles; consecutive batches on different lanes and HIP
    streams (what the sweep and bench.py do) give the results of the single-stream execution."""
    sd = synthetic_state_dict(**TINY)
    model, _ = clip.load(sd, max_batch=16)
    xs = [synthetic_images(5, seed=s).to(cuda) for s in range(4)]
    want = [model.encode_image(x, normalize=True, out_dtype=torch.float32).clone() for x in xs]
    streams = [torch.cuda.Stream(cuda) for _ in range(2)]
    torch.cuda.synchronize()
    got = []
    for i, x in enumerate(xs):
        model.visual.lane = i % 2
        with torch.cuda.stream(streams[i % 2]):
            got.append(model.encode_image(x, normalize=True, out_dtype=torch.float32))
    model.visual.lane = 0
    torch.cuda.synchronize()
    assert len(model.visual._lanes) == 2 and len({h.value for h, _ in model.visual._lanes.values()}) == 2
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    model.visual.close()
    assert not model.visual._lanes


def _clip_like_statistics(sd, width=768, seed=5):
    """Push the benign synthetic weights towards what a trained CLIP ViT looks like numerically: LayerNorm
    gains spread over more than a decade with a few large ones, sizeable LayerNorm shifts, 'massive
    activation' channels in the residual stream (a few channels of the class / positional embedding and of the
    residual-branch biases tens of times larger than the rest), sharper attention logits."""
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    big = torch.randperm(width, generator=g)[:4]
    for k in sd:
        if k.endswith(('ln_1.weight', 'ln_2.weight', 'ln_pre.weight', 'ln_post.weight')):
            gain = torch.exp(torch.empty(width).uniform_(-1.6, 1.4, generator=g))  # 0.2 .. 4
            gain[torch.randperm(width, generator=g)[:3]] *= 8.0
            sd[k] = gain
        elif k.endswith(('ln_1.bias', 'ln_2.bias', 'ln_pre.bias', 'ln_post.bias')):
            sd[k] = torch.randn(width, generator=g) * 0.5
        elif k.endswith(('out_proj.bias', 'c_proj.bias')):
            sd[k][big] += torch.tensor([6.0, -5.0, 4.0, -7.0])  # outlier channels that grow layer by layer
        elif k.endswith('in_proj_weight'):
            sd[k][:2 * width] *= 1.8  # q and k: sharper softmax
    sd['visual.class_embedding'][big] *= 40.0
    sd['visual.positional_embedding'][:, big] *= 25.0
    return sd


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 2e-2)])
def test_encode_image_clip_like_weight_statistics(cuda, dtype, tol):
    """The 16-bit residual stream, the LayerNorm folded into the consuming GEMMs (gamma in W, row statistics
    applied in the epilogue) and the GEMM-to-GEMM hand-off of (sum x, sum x^2) under hostile statistics:
    residual rows whose mean and variance are dominated by a few channels, LayerNorm gains from 0.2 to 30.
    Against the fp32 oracle, at the north-star tolerance."""
    sd = _clip_like_statistics(synthetic_state_dict())
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=48)
    x = synthetic_images(48, seed=321)
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x))
    out = model.encode_image(x.to(cuda), normalize=True, out_dtype=torch.float32)
    assert torch.isfinite(out).all()
    _check(out, ref, tol, tol)
    small = model.encode_image(x[:5].to(cuda), normalize=True, out_dtype=torch.float32)  # non-persistent kernels
    _check(small, ref[:5], tol, tol)


def test_two_handles_on_two_host_threads(cuda):
    """A handle is not thread-safe, but distinct handles may be driven from distinct host threads (ctypes drops
    the GIL inside the library): two models with different weights encode different batches on their own HIP
    streams at the same time, twenty rounds each — every result equals the one computed alone."""
    import threading
    sds = [synthetic_state_dict(seed=s, **TINY) for s in (1, 2)]
    models = [clip.load(sd, max_batch=40)[0] for sd in sds]
    xs = [synthetic_images(n, seed=50 + n).to(cuda) for n in (37, 64)]
    alone = [m.encode_image(x, normalize=True, out_dtype=torch.float32).clone() for m, x in zip(models, xs)]
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            stream = torch.cuda.Stream(cuda)
            with torch.cuda.stream(stream):
                for _ in range(20):
                    out = models[i].encode_image(xs[i], normalize=True, out_dtype=torch.float32)
                    stream.synchronize()
                    if not torch.equal(out, alone[i]):
                        errors.append(f'thread {i}: result differs')
                        return
        except Exception as e:  # noqa: BLE001
            errors.append(f'thread {i}: {e!r}')

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize('in_dtype', [torch.float16, torch.float32])
def test_objects_conv1_from_padded_buffer_equals_im2col(cuda, in_dtype):
    """Objects mode's conv1 (patch 32, stride 16, padding 15: overlapping patches, zero border) gathered by the
    GEMM from a zero-padded 16-bit copy of the batch vs the same GEMM on the im2col matrix: the same operand
    bits in the same order => bit-identical features; fp32 inputs (cast in the pad pass) included.  70 crops =
    13 720 patch rows: ragged last tile, several tiles per persistent block."""
    sd = synthetic_state_dict()

    def make(patch_direct):
        model, _ = clip.load(sd, max_batch=70)
        v = model.visual
        v.positional_embedding = v.interpolate_positional_embedding((14, 14))
        v.grid = 14
        v.conv1.stride = (16, 16)
        v.conv1.padding = (15, 15)
        v.object_stream = True
        v.set_option('patch_direct', patch_direct)
        return v

    a, b = make(1), make(0)
    x = synthetic_images(70, seed=12).to(in_dtype).to(cuda)
    masks = (torch.rand(70, 1, 14, 14, generator=torch.Generator().manual_seed(3)) < 0.6).half().to(cuda)
    ya = a(x, masks, normalize=True, out_dtype=torch.float32)
    yb = b(x, masks, normalize=True, out_dtype=torch.float32)
    assert torch.equal(ya, yb)
    a.profile(True)
    a(x, masks)
    names = {p_['name'] for p_ in a.profile_read()}
    a.profile(False)
    assert 'im2col' not in names and 'pad_nchw' in names and 'gemm_conv1' in names
    sd2 = dict(sd)
    sd2['visual.positional_embedding'] = a.positional_embedding.detach().cpu().float()
    ref = l2_normalize(encode_objects_ref(sd2, ViTConfig(stride=16, padding=15), x[:4].cpu().float(), masks[:4].cpu().float()))
    _check(ya[:4], ref, 1e-3, 1e-3)


def test_objects_crops_written_into_the_padded_batch(cuda):
    """objects mode's crops go STRAIGHT into the zero-padded 16-bit batch conv1 gathers from
    (OAKE_LAYOUT_PADDED: csrc/resample.hip resample_v4p_kernel, `visual.crop_resize_normalize_batch` hands out a strided
    view of a per-lane pool) and the library's pad pass is not run [REF oadp/oake/objects.py:116-127,298-301].
    (a) the view's values == the dense crops, bit for bit, for boxes of every kind (in / across every border, tiny, one
    exactly 224 wide: no resampling); (b) every border element of the pool is zero, also after a SECOND call with fewer
    crops and other boxes reused the pool; (c) features via the view == via the dense tensor (the same padded operand
    bits reach conv1), and the profile shows no pad_nchw; (d) a slice of 3 crops (too few rows for the persistent GEMM:
    the library unpads and takes the im2col route) == the dense path too."""
    import numpy as np
    sd = synthetic_state_dict()

    def make():
        model, _ = clip.load(sd, max_batch=70)
        v = model.visual
        v.positional_embedding = v.interpolate_positional_embedding((14, 14))
        v.grid = 14
        v.conv1.stride = (16, 16)
        v.conv1.padding = (15, 15)
        v.object_stream = True
        return v

    a, b = make(), make()
    b.padded_crops = False
    rng = np.random.default_rng(4)
    imgs = [torch.from_numpy(rng.integers(0, 256, size=(hh, ww, 3), dtype=np.uint8)).to(cuda)
            for ww, hh in ((640, 480), (333, 517), (224, 224))]

    def boxes_for(seed, per):
        r = np.random.default_rng(seed)
        out = []
        for im in imgs:
            hh, ww = im.shape[:2]
            bs = [(0, 0, ww, hh), (-20.5, -10.5, ww * 0.6, hh * 0.7), (ww * 0.3, hh * 0.2, ww + 33.2, hh + 5.5),
                  (ww * 0.45, hh * 0.45, ww * 0.45 + 4.2, hh * 0.45 + 9.7), (0, 0, min(ww, 224), min(hh, 224))]
            for _ in range(per):
                x1, y1, side = r.uniform(-30, ww * 0.8), r.uniform(-30, hh * 0.8), r.uniform(4, max(ww, hh))
                bs.append((x1, y1, x1 + side, y1 + side))
            out.append([tuple(float(t) for t in bb) for bb in bs])
        return out

    for seed, per in ((1, 18), (2, 9)):  # 69 crops, then 42 into the same pool
        boxes = boxes_for(seed, per)
        va = a.crop_resize_normalize_batch(imgs, boxes, out_dtype=torch.float16)
        vb = b.crop_resize_normalize_batch(imgs, boxes, out_dtype=torch.float16)
        n = vb.shape[0]
        assert va.shape == vb.shape and not va.is_contiguous() and vb.is_contiguous()
        assert torch.equal(va, vb)
        pool, pad, hp, ws = next(iter(a._pad_pools.values()))
        assert (pad, hp, ws) == (15, 254, 256) and pool.shape[0] >= n
        border = pool.clone()
        border[:, :, pad:pad + 224, pad:pad + 224] = 0
        assert not border.any()  # (rows past n keep the previous call's interiors: only the BORDER must be zero)
        masks = (torch.rand(n, 1, 14, 14, generator=torch.Generator().manual_seed(seed)) < 0.6).half().to(cuda)
        a.profile(True)
        ya = a(va, masks, normalize=True, out_dtype=torch.float32)
        names = {p_['name'] for p_ in a.profile_read() if p_['launches'] > 0}
        a.profile(False)
        assert 'pad_nchw' not in names and 'im2col' not in names and 'gemm_conv1' in names, names
        yb = b(vb, masks, normalize=True, out_dtype=torch.float32)
        assert torch.equal(ya, yb)
        # dim-0 slices of the view stay views of the pool; a handful of crops takes the unpad + im2col route
        assert torch.equal(a(va[40:], masks[40:], normalize=True, out_dtype=torch.float32),
                           b(vb[40:], masks[40:], normalize=True, out_dtype=torch.float32))
        a.profile(True)
        y3 = a(va[5:8], masks[5:8], normalize=True, out_dtype=torch.float32)
        names = {p_['name'] for p_ in a.profile_read() if p_['launches'] > 0}
        a.profile(False)
        assert 'unpad_nchw' in names and 'im2col' in names, names
        assert torch.equal(y3, b(vb[5:8], masks[5:8], normalize=True, out_dtype=torch.float32))
    # a dense copy of the view is an ordinary tensor again
    assert torch.equal(a(va.contiguous(), masks, normalize=True, out_dtype=torch.float32), ya)
    # a narrow tower (width 128: conv1 does not run the persistent GEMM at any pass size): EVERY pass of a padded-layout
    # call is unpadded into the library's own scratch — found by tests/fuzz_pipeline.py when that copy went into a
    # workspace buffer that is smaller than the crops in such a geometry
    sdt = synthetic_state_dict(**TINY)
    mt, _, _ = _objects_model(sdt, TINY, max_batch=16)
    md, _, _ = _objects_model(sdt, TINY, max_batch=16)
    md.visual.padded_crops = False
    boxes = boxes_for(3, 12)  # 51 crops: several passes
    vt = mt.visual.crop_resize_normalize_batch(imgs, boxes, out_dtype=torch.float16)
    vd = md.visual.crop_resize_normalize_batch(imgs, boxes, out_dtype=torch.float16)
    assert not vt.is_contiguous() and torch.equal(vt, vd)
    mk = (torch.rand(vt.shape[0], 1, 14, 14, generator=torch.Generator().manual_seed(9)) < 0.5).half().to(cuda)
    assert torch.equal(mt.visual(vt, mk, normalize=True, out_dtype=torch.float32),
                       md.visual(vd, mk, normalize=True, out_dtype=torch.float32))


def test_cu_count_option_and_masked_stream(cuda, lib):
    """OAKE_OPT_CU_COUNT sizes the persistent grids for a CU-masked stream (round 3's half-chip lanes experiment,
    oadp_amd/cumask.py): fewer, longer-running blocks walk the same tiles — bit-identical features — and a stream
    created over the first half of the mask bits really runs on half of the CUs, half of EVERY XCD."""
    import ctypes as C
    from oadp_amd import cumask
    sd = synthetic_state_dict()
    full, _ = clip.load(sd, max_batch=64)
    half, _ = clip.load(sd, max_batch=64)
    half.visual.set_option('cu_count', 128)
    x = synthetic_images(64, seed=91).half().to(cuda)  # 3200 rows: the persistent kernels
    want = full.encode_image(x, normalize=True, out_dtype=torch.float32)
    ncu = torch.cuda.get_device_properties(cuda).multi_processor_count
    masks = cumask.half_masks(ncu, 'halves')
    stream = torch.cuda.ExternalStream(cumask.create_masked_stream(masks[0]), device=cuda)
    with torch.cuda.stream(stream):
        got = half.encode_image(x, normalize=True, out_dtype=torch.float32)
    stream.synchronize()
    assert torch.equal(got, want)
    out = torch.zeros((512, 2), dtype=torch.int32, device=cuda)
    assert lib.oake_debug_cu_census(out.data_ptr(), 512, 200, C.c_void_p(stream.cuda_stream)) == 0
    stream.synchronize()
    o = out.cpu().numpy().astype('uint32')
    seen = {(int(xcc) & 15, (int(hw) >> 13) & 7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 15) for xcc, hw in o}
    per_xcc = {}
    for k in seen:
        per_xcc[k[0]] = per_xcc.get(k[0], 0) + 1
    assert len(seen) == ncu // 2 and len(per_xcc) == 8 and set(per_xcc.values()) == {ncu // 16}, per_xcc


def test_profiler_counts_stamped_and_seen_launches(cuda):
    sd = synthetic_state_dict(**TINY)
    model, _ = clip.load(sd, max_batch=8)
    x = synthetic_images(4, seed=3).to(cuda)
    model.encode_image(x)
    v = model.visual
    v.profile(True)
    model.encode_image(x)
    full = v.profile_read()
    assert full and all(p['launches'] == p['seen'] >= 1 and p['total_ms'] > 0 for p in full)
    n = sum(p['seen'] for p in full)
    v.profile(3)  # every third launch stamped
    model.encode_image(x)
    sparse = v.profile_read()
    v.profile(False)
    assert sum(p['seen'] for p in sparse) == n and sum(p['launches'] for p in sparse) == (n + 2) // 3


@pytest.mark.parametrize('dtype,tol', [(torch.float16, 1e-3), (torch.bfloat16, 2e-2)])
def test_encode_image_fused_attention_out_proj(cuda, dtype, tol):
    """The one-kernel attention + out_proj + residual (csrc/attn_out.hip; measured slower, so it lives in the LAB
    library only: docs/history/round4.md item 4) against the oracle, against the two-launch form (fuse_attn_out = 0: same
    arithmetic up to the summation order of out_proj's K dimension), and that it really is the path taken (profile
    slot names).  The product library refuses the option."""
    from oadp_amd import _lib
    sd = synthetic_state_dict()
    prod, _ = clip.load(sd, compute_dtype=dtype, max_batch=4)
    prod.encode_image(synthetic_images(2, seed=1).to(cuda))
    with pytest.raises(_lib.OakeError, match='liboake_hip_lab'):
        prod.visual.set_option('fuse_attn_out', 1)
    prod.visual.set_option('fuse_attn_out', 0)  # switching it OFF is always valid
    model, _ = clip.load(sd, compute_dtype=dtype, max_batch=48, lib=_lib.load_lab())
    x = synthetic_images(45, seed=145)
    ref = l2_normalize(encode_image_ref(sd, ViTConfig(), x))
    v = model.visual
    xg = x.to(cuda)
    model.encode_image(xg[:2])  # creates the handle
    v.set_option('fuse_qkv_attn', 0)  # (the production default fuses the OTHER side of the attention: csrc/qkv_attn.hip)
    v.set_option('fuse_attn_out', 1)
    v.profile(True)
    fused = model.encode_image(xg, normalize=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    names = {p['name']: p['launches'] for p in v.profile_read() if p['launches'] > 0}
    v.profile(False)
    assert names.get('attn_out') == 11 and 'attention' not in names and 'gemm_out_proj' not in names, names
    _check(fused, ref, tol, tol)
    v.set_option('fuse_attn_out', 0)
    v.profile(True)
    plain = model.encode_image(xg, normalize=True, out_dtype=torch.float32)
    torch.cuda.synchronize()
    names = {p['name']: p['launches'] for p in v.profile_read() if p['launches'] > 0}
    v.profile(False)
    assert 'attn_out' not in names and names.get('attention') == 11 and names.get('gemm_out_proj') == 11, names
    _check(plain, ref, tol, tol)
    assert (fused - plain).abs().max().item() <= tol
    # per-image results do not depend on the batch composition (one workgroup per image)
    v.set_option('fuse_attn_out', 1)
    again = model.encode_image(xg.flip(0), normalize=True, out_dtype=torch.float32).flip(0)
    assert torch.equal(again, fused)
