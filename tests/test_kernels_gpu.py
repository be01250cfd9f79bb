"""Kernel-level parity on the GPU, through the C ABI debug entry points (include/oake_hip.h).

Each HIP kernel is compared with a plain PyTorch fp32 reference of the same op computed from the
SAME 16-bit-rounded operands, so the tolerance only has to cover fp32 accumulation order and the
16-bit rounding of outputs.  Inputs are asymmetric random (a transposed C-write cannot pass).
"""
import ctypes as C

import pytest
import torch

from oadp_amd import _lib

pytestmark = pytest.mark.gpu

DT = {torch.float16: _lib.OAKE_F16, torch.bfloat16: _lib.OAKE_BF16}
PROD_GEMM = (-2, -1, 0, 4, 5, 13)  # tile configurations of the production library; the others live in liboake_hip_lab.so


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('m,n,k', [(128, 128, 64), (256, 384, 128), (1350, 768, 768),
                                   (50, 2304, 768), (77, 132, 3072), (12800, 768, 3072)])
def test_gemm(lib, cuda, dtype, m, n, k):
    _gemm_case(lib, cuda, dtype, m, n, k)


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13])
@pytest.mark.parametrize('m,n,k', [(320, 256, 64), (1350, 768, 768), (50, 2304, 192), (333, 132, 3072), (12800, 768, 128)])
def test_gemm_tile_configs(lib, lab, cuda, variant, m, n, k):
    """Every tile configuration (csrc/gemm.hip) on ragged M/N edges."""
    lib = lib if variant in PROD_GEMM else lab
    lib.oake_debug_set_gemm_variant(variant)
    try:
        _gemm_case(lib, cuda, torch.float16, m, n, k)
    finally:
        lib.oake_debug_set_gemm_variant(-1)


@pytest.mark.parametrize('variant', [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize('gelu', [0, 1])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('m,n,k', [(320, 256, 64), (1350, 768, 768), (50, 2304, 192), (333, 136, 512),
                                   (12800, 3072, 128), (12800, 3072, 768), (12800, 2048, 192), (5000, 1288, 256)])
def test_gemm_16bit_epilogues(lib, lab, cuda, variant, gelu, dtype, m, n, k):
    """QKV / c_fc epilogues: bias (+QuickGELU) and the paired-column 16-byte stores."""
    lib = lib if variant in PROD_GEMM else lab
    g = torch.Generator(device='cpu').manual_seed(m + n + k + gelu)
    a = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(cuda)
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).to(dtype).to(cuda)
    bias = torch.randn(n, generator=g).to(cuda)
    c = torch.full((m, n), float('nan'), dtype=dtype, device=cuda)
    lib.oake_debug_set_gemm_variant(variant)
    try:
        rc = lib.oake_debug_gemm16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k,
                                   DT[dtype], gelu, _stream())
        assert rc == 0
        torch.cuda.synchronize()
    finally:
        lib.oake_debug_set_gemm_variant(-1)
    ref = a.float() @ w.float().t() + bias
    if gelu:
        ref = ref * torch.sigmoid(1.702 * ref)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    torch.testing.assert_close(c.float(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize('variant', [0, 1, 4, 5, 6, 8, 12, 13])
@pytest.mark.parametrize('gelu', [0, 1])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('m,n,k', [(320, 256, 192), (1350, 768, 768), (50, 2304, 768), (333, 136, 512),
                                   (12800, 1024, 256), (12800, 3072, 768), (25600, 1536, 192), (5003, 1288, 448),
                                   (1000, 520, 768)])
def test_gemm_layernorm_folded(lib, lab, cuda, variant, gelu, dtype, m, n, k):
    """LayerNorm folded into the consuming GEMM (16-bit residual stream): gamma in W, beta in the
    bias, per-row (rstd, -mean*rstd) applied in the epilogue == GEMM(LayerNorm(x)) in fp32."""
    lib = lib if variant in PROD_GEMM else lab
    g = torch.Generator(device='cpu').manual_seed(m + n + k + gelu)
    x = torch.randn(m, k, generator=g) * 1.5 + 0.3
    x[:, 5] *= 12.0          # CLIP's residual stream has a few large-magnitude channels
    x[:, k // 2] += 7.0
    x = x.to(dtype).to(cuda)
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(k, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(k, generator=g)).to(cuda)
    bias = torch.randn(n, generator=g).to(cuda)
    c = torch.full((m, n), float('nan'), dtype=dtype, device=cuda)
    lib.oake_debug_set_gemm_variant(variant)
    try:
        rc = lib.oake_debug_ln_gemm16(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      bias.data_ptr(), c.data_ptr(), m, n, k, DT[dtype], gelu, _stream())
        assert rc == 0
        torch.cuda.synchronize()
    finally:
        lib.oake_debug_set_gemm_variant(-1)
    ref = torch.nn.functional.layer_norm(x.float(), (k,), gamma, beta, 1e-5) @ w.t() + bias
    if gelu:
        ref = ref * torch.sigmoid(1.702 * ref)
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    torch.testing.assert_close(c.float(), ref, rtol=tol, atol=tol)


@pytest.mark.parametrize('gelu', [0, 1])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('m,n,k', [(12800, 3072, 768), (25600, 768, 768), (5003, 1288, 448), (640, 520, 768),
                                   (12800, 768, 3072), (5003, 512, 448)])
def test_gemm_320_row_tile_matches_the_160_row_kernel_bit_for_bit(lib, cuda, gelu, dtype, m, n, k):
    """c_fc runs on the 320 x 256 kernel (variant 13) when its tiles fill the chip and on the 160 x 256 one (variant 4)
    otherwise — a choice that depends on the row count, so the two must agree in every bit: same MFMA order, the same
    slot-order sum of the row statistics, the same epilogue arithmetic."""
    g = torch.Generator(device='cpu').manual_seed(m + n + k + gelu)
    x = torch.randn(m, k, generator=g) * 1.5 + 0.3
    x[:, 5] *= 12.0
    x = x.to(dtype).to(cuda)
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(k, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(k, generator=g)).to(cuda)
    bias = torch.randn(n, generator=g).to(cuda)
    if k <= 1024:  # (LayerNorm-folded: the row-statistics slots cover a residual width of 16 x 64 columns)
        outs = []
        for variant in (4, 13):
            c = torch.full((m, n), float('nan'), dtype=dtype, device=cuda)
            lib.oake_debug_set_gemm_variant(variant)
            try:
                assert lib.oake_debug_ln_gemm16(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
                                                c.data_ptr(), m, n, k, DT[dtype], gelu, _stream()) == 0
                torch.cuda.synchronize()
            finally:
                lib.oake_debug_set_gemm_variant(-1)
            outs.append(c)
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    a = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(cuda)
    w16 = w.to(dtype)
    outs = []
    for variant in (4, 13):
        c = torch.full((m, n), float('nan'), dtype=dtype, device=cuda)
        lib.oake_debug_set_gemm_variant(variant)
        try:
            assert lib.oake_debug_gemm16(a.data_ptr(), w16.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k, DT[dtype], gelu,
                                         _stream()) == 0
            torch.cuda.synchronize()
        finally:
            lib.oake_debug_set_gemm_variant(-1)
        outs.append(c)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    # the residual epilogue (out_proj / c_proj where a step has 25 600 rows): x and the row-statistics slices
    if n % 64 == 0 and n <= 1024:  # (the row-statistics slots: 16 slices of 64 columns)
        x0 = torch.randn(m, n, generator=g).to(dtype).to(cuda)
        outs = []
        for variant in (4, 13):
            x = x0.clone()
            part = torch.zeros((m, 16, 2), device=cuda)
            lib.oake_debug_set_gemm_variant(variant)
            try:
                assert lib.oake_debug_gemm_resid16(a.data_ptr(), w16.data_ptr(), bias.data_ptr(), x.data_ptr(), part.data_ptr(),
                                                   m, n, k, DT[dtype], _stream()) == 0
                torch.cuda.synchronize()
            finally:
                lib.oake_debug_set_gemm_variant(-1)
            outs.append((x, part))
        assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))
        assert torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32))


@pytest.mark.parametrize('variant', [-1, 10, 13])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('m,n,k', [(320, 256, 192), (1350, 768, 768), (2250, 768, 3072), (12800, 512, 256),
                                   (333, 136, 512), (50, 768, 768), (41000, 768, 192), (25600, 768, 768)])
def test_gemm_residual_epilogue(lib, lab, cuda, dtype, m, n, k, variant):
    """x += A W^T + b in the 16-bit residual stream (out_proj / c_proj), ragged tiles included, plus the
    per-slice row sums the persistent kernel leaves for the next GEMM's LayerNorm.  The last two shapes give
    every persistent block several tiles with different bias blocks (771 / 480 tiles on 256 CUs; K = 192 is
    the shortest K loop the kernel takes): the staging of a tile's epilogue constants must not overtake the
    previous tile's epilogue."""
    lib = lib if variant in PROD_GEMM else lab
    g = torch.Generator(device='cpu').manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(cuda)
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).to(dtype).to(cuda)
    bias = torch.randn(n, generator=g).to(cuda)
    x0 = torch.randn(m, n, generator=g).to(dtype).to(cuda)
    x = x0.clone()
    part = torch.full((m, 16, 2), float('nan'), device=cuda)
    lib.oake_debug_set_gemm_variant(variant)  # -1: automatic (two long phases per K-tile); 10: four short phases
    try:
        rc = lib.oake_debug_gemm_resid16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(),
                                         part.data_ptr(), m, n, k, DT[dtype], _stream())
        assert rc == 0
        torch.cuda.synchronize()
    finally:
        lib.oake_debug_set_gemm_variant(-1)
    ref = x0.float() + a.float() @ w.float().t() + bias
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    torch.testing.assert_close(x.float(), ref, rtol=tol, atol=tol)
    if m > 1024 and n >= 256 and n % 64 == 0:  # persistent kernel: row statistics were produced
        got = part[:, :n // 64]
        assert torch.isfinite(got).all()
        torch.testing.assert_close(got[..., 0], ref.reshape(m, n // 64, 64).sum(-1), rtol=2e-3, atol=2e-2)
        torch.testing.assert_close(got[..., 1], (ref * ref).reshape(m, n // 64, 64).sum(-1), rtol=2e-3, atol=2e-2)


def _gemm_case(lib, cuda, dtype, m, n, k):
    g = torch.Generator(device='cpu').manual_seed(m * 7 + n * 3 + k)
    a = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(cuda)
    w = (torch.randn(n, k, generator=g) * (k ** -0.5)).to(dtype).to(cuda)
    bias = torch.randn(n, generator=g).to(cuda)
    c = torch.full((m, n), float('nan'), device=cuda)
    rc = lib.oake_debug_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), m, n, k,
                             DT[dtype], _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + bias
    torch.testing.assert_close(c, ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('rows,c', [(1, 768), (50, 768), (12800, 768), (7, 128), (33, 1024)])
@pytest.mark.parametrize('x16', [False, True])
def test_layernorm(lib, cuda, dtype, rows, c, x16):
    g = torch.Generator(device='cpu').manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 3 + 0.7).to(cuda)
    if x16:
        x = x.to(dtype)
    gamma = (1 + 0.1 * torch.randn(c, generator=g)).to(cuda)
    beta = (0.1 * torch.randn(c, generator=g)).to(cuda)
    y = torch.zeros(rows, c, dtype=dtype, device=cuda)
    rc = lib.oake_debug_layernorm(x.data_ptr(), DT[dtype] if x16 else _lib.OAKE_F32, gamma.data_ptr(),
                                  beta.data_ptr(), y.data_ptr(), rows, c, DT[dtype], _stream())
    assert rc == 0
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.float(), (c,), gamma, beta, 1e-5)
    tol = 2e-3 if dtype == torch.float16 else 2e-2
    torch.testing.assert_close(y.float(), ref, rtol=tol, atol=tol)


def _attention_ref(qkv, n, l, heads):
    c = heads * 64
    q, k, v = qkv.float().view(n, l, 3, heads, 64).permute(2, 0, 3, 1, 4)  # [n, h, l, 64]
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)  # q is pre-scaled
    return (p @ v).permute(0, 2, 1, 3).reshape(n * l, c)


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('form', ['triple', 'quad'])
@pytest.mark.parametrize('n,l,heads', [(3, 50, 12), (1, 50, 12), (2, 50, 12), (256, 50, 12), (7, 50, 3), (100, 50, 12),
                                       (5, 53, 4), (4, 17, 3), (3, 1, 3), (9, 33, 12), (770, 50, 12), (5, 49, 4),
                                       (1030, 50, 12), (6, 48, 3)])
def test_ln_qkv_attention_fused(lib, lab, cuda, dtype, n, l, heads, form):
    """csrc/qkv_attn.hip: ln_1 folded into attn.in_proj + softmax(q k^T) v as ONE persistent kernel (a tile = three
    images x one head; q | k | v go from the accumulators through LDS into the attention) against the same chain in
    fp32 torch from the same 16-bit x: LayerNorm -> in_proj -> ROUNDED to 16 bits (as the two-launch form stores it) ->
    attention.  Shapes: full groups, a ragged last group (1 and 2 images), several tiles per block with different
    heads (256 x 12 = 1032 tiles; 770 images = 3084 tiles: 12 per block), other sequence lengths and head counts.
    form 'quad': the same contract through csrc/qkv_attn_obj.hip's four-images-per-208-row-tile form (l <= 50; odd l: the
    images' rows start at odd LDS rows; 1030 images: a last group of two)."""
    if form == 'quad' and l > 50:
        pytest.skip('four images of more than 50 tokens do not fit the 208-row tile')
    # (the three-image form lost its A/B and is in the lab build only; the production library answers UNSUPPORTED)
    entry = lib.oake_debug_ln_qkv_attn_quad if form == 'quad' else lab.oake_debug_ln_qkv_attn
    c = heads * 64
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + l * 10 + heads)
    x = torch.randn(n * l, c, generator=g) * 1.5 + 0.3
    x[:, 5] *= 12.0  # CLIP's residual stream has a few large-magnitude channels
    x = x.to(dtype).to(cuda)
    w = torch.randn(3 * c, c, generator=g) * (c ** -0.5)
    w[:c] *= 0.35  # pre-scaled q rows: scores ~ N(0, 2.8^2), a peaky softmax
    w = w.to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(c, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(c, generator=g)).to(cuda)
    bias = (0.5 * torch.randn(3 * c, generator=g)).to(cuda)
    out = torch.full((n * l + 3, c), 7.0, dtype=dtype, device=cuda)  # guard rows behind the last image
    rc = entry(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
               out.data_ptr(), n, l, heads, DT[dtype], None, 1, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    # (a) the attention of the 16-bit q | k | v the two-launch form stores (oake_debug_ln_gemm16: same folded weights,
    # same K order of the fp32 accumulation -> the same rounded values), in fp32 torch: attention-kernel tolerance
    qkv16 = torch.empty(n * l, 3 * c, dtype=dtype, device=cuda)
    assert lib.oake_debug_ln_gemm16(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
                                    qkv16.data_ptr(), n * l, 3 * c, c, DT[dtype], 0, _stream()) == 0
    torch.cuda.synchronize()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    torch.testing.assert_close(out[:n * l].float(), _attention_ref(qkv16, n, l, heads), rtol=tol, atol=tol)
    # (b) the whole chain in fp32 torch: the 16-bit rounding of q / k / v (1 ulp either way) moves a peaky softmax
    qkv = (torch.nn.functional.layer_norm(x.float(), (c,), gamma, beta, 1e-5) @ w.t() + bias).to(dtype)
    ref = _attention_ref(qkv, n, l, heads)
    err = (out[:n * l].float() - ref).abs()
    assert err.max().item() < (0.05 if dtype == torch.float16 else 0.4) and err.mean().item() < (1.5e-3 if dtype == torch.float16 else 1.5e-2)
    assert torch.equal(out[n * l:], torch.full((3, c), 7.0, dtype=dtype, device=cuda))  # nothing written past the rows


def _objects_attention_ref(qkv, n, l, heads, mask):
    """The two attentions of objects mode from q | k | v rows [n*l + n, 3C] (token rows, then one object-token row per
    crop): tokens over their crop's tokens; the object token over its crop's PATCH rows (-100 * mask) and itself
    [REF oadp/oake/objects.py:223-247]."""
    c = heads * 64
    x, y = qkv[:n * l].float(), qkv[n * l:].float()
    q, k, v = x.view(n, l, 3, heads, 64).permute(2, 0, 3, 1, 4)                 # [n, h, l, 64]
    out_x = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n * l, c)
    qy, ky, vy = y.view(n, 1, 3, heads, 64).permute(2, 0, 3, 1, 4)              # [n, h, 1, 64]
    keys, vals = torch.cat([k[:, :, 1:], ky], 2), torch.cat([v[:, :, 1:], vy], 2)
    bias = torch.cat([-100.0 * mask.float(), torch.zeros(n, 1, device=mask.device)], 1)[:, None, None, :]
    out_y = (torch.softmax(qy @ keys.transpose(-1, -2) + bias, dim=-1) @ vals).permute(0, 2, 1, 3).reshape(n, c)
    return torch.cat([out_x, out_y])


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('mask_dtype', [torch.float16, torch.float32])
@pytest.mark.parametrize('n,l,heads', [(1, 197, 12), (3, 197, 12), (128, 197, 12), (5, 197, 3), (2, 192, 4), (40, 199, 5),
                                       (30, 193, 12), (300, 197, 12)])
def test_ln_qkv_attention_objects_fused(lib, cuda, dtype, mask_dtype, n, l, heads):
    """csrc/qkv_attn_obj.hip: objects mode's ln_1 + in_proj + BOTH attentions (a crop's tokens; its object token with the
    mask bias) as ONE persistent kernel, a tile = (crop, head) of 208 rows, against fp32 torch on the 16-bit q | k | v the
    two-launch form stores (oake_debug_ln_gemm16).  128 crops = 1536 tiles = six per block; 300 crops: ragged rounds;
    l = 192 / 193 / 199: the ends of the range the form takes (13 key tiles, two regions of l + 1 rows in one ring
    slot); masks: random, one crop all foreground, one all
    background."""
    if n * l * heads * 64 > 30_000_000:
        heads = 12
    c = heads * 64
    m = n * l + n
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + l * 10 + heads)
    x = torch.randn(m, c, generator=g) * 1.5 + 0.3
    x[:, 7] *= 12.0
    x = x.to(dtype).to(cuda)
    w = torch.randn(3 * c, c, generator=g) * (c ** -0.5)
    w[:c] *= 0.35
    w = w.to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(c, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(c, generator=g)).to(cuda)
    bias = (0.5 * torch.randn(3 * c, generator=g)).to(cuda)
    mask = (torch.rand(n, l - 1, generator=g) < 0.4).float()
    mask[0] = 0
    if n > 1:
        mask[1] = 1
    md = mask.to(mask_dtype).to(cuda)
    out = torch.full((m + 3, c), 7.0, dtype=dtype, device=cuda)
    rc = lib.oake_debug_ln_qkv_attn_obj(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
                                        md.data_ptr(), _lib.OAKE_F16 if mask_dtype == torch.float16 else _lib.OAKE_F32,
                                        out.data_ptr(), n, l, heads, DT[dtype], None, 1, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    qkv16 = torch.empty(m, 3 * c, dtype=dtype, device=cuda)
    assert lib.oake_debug_ln_gemm16(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(),
                                    qkv16.data_ptr(), m, 3 * c, c, DT[dtype], 0, _stream()) == 0
    torch.cuda.synchronize()
    ref = _objects_attention_ref(qkv16, n, l, heads, mask.to(cuda))
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    torch.testing.assert_close(out[:n * l].float(), ref[:n * l], rtol=tol, atol=tol)   # the token stream
    torch.testing.assert_close(out[n * l:m].float(), ref[n * l:], rtol=tol, atol=tol)  # the object tokens
    assert torch.equal(out[m:], torch.full((3, c), 7.0, dtype=dtype, device=cuda))


def test_ln_qkv_attention_refuses_long_sequences(lib, lab, cuda):
    z = torch.zeros(64, device=cuda)
    for bad_l in (191, 200, 207, 50):  # objects form: 192 <= l <= 199 only
        assert lib.oake_debug_ln_qkv_attn_obj(*([z.data_ptr()] * 6), _lib.OAKE_F16, z.data_ptr(), 1, bad_l, 12,
                                              _lib.OAKE_F16, None, 1, _stream()) == _lib.OAKE_ERR_UNSUPPORTED
    args = [z.data_ptr()] * 6
    assert lab.oake_debug_ln_qkv_attn(*args, 1, 54, 12, _lib.OAKE_F16, None, 1, _stream()) == _lib.OAKE_ERR_UNSUPPORTED
    assert lab.oake_debug_ln_qkv_attn(*args, 1, 197, 12, _lib.OAKE_F16, None, 1, _stream()) == _lib.OAKE_ERR_UNSUPPORTED
    # the production library does not carry the three-image form at all
    assert lib.oake_debug_ln_qkv_attn(*args, 3, 50, 12, _lib.OAKE_F16, None, 1, _stream()) == _lib.OAKE_ERR_UNSUPPORTED
    assert lib.oake_debug_ln_qkv_attn_quad(*args, 1, 51, 12, _lib.OAKE_F16, None, 1, _stream()) == _lib.OAKE_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize('walk', [1, 2, 3, 4, 6, 12, 5])
@pytest.mark.parametrize('form,n,l,heads', [('quad', 256, 50, 12), ('quad', 430, 50, 12), ('quad', 37, 49, 6),
                                            ('obj', 128, 197, 12), ('obj', 75, 197, 12), ('obj', 9, 193, 6)])
def test_ln_qkv_attention_tile_walk_is_placement_only(lib, cuda, walk, form, n, l, heads):
    """OAKE_OPT_QKV_WALK (csrc/qkv_attn_obj.hip::walk_decode): the order in which an XCD's blocks visit the (group, head)
    tiles — head blocks of `walk` heads x group blocks of 32 / walk groups — is a bijection of the tile set for full and
    ragged group counts (430 images = 108 groups: XCD ranges cut group blocks; 75 crops; 6 heads: walk 4, 12 and 5 do
    not divide and fall back to group-major), so every output row is BIT-identical to the group-major walk's."""
    c = heads * 64
    m = n * l + (n if form == 'obj' else 0)
    g = torch.Generator(device='cpu').manual_seed(n * 131 + l + heads)
    x = (torch.randn(m, c, generator=g) * 1.5 + 0.3).half().to(cuda)
    w = (torch.randn(3 * c, c, generator=g) * (c ** -0.5)).to(cuda)
    gamma = (1.0 + 0.3 * torch.randn(c, generator=g)).to(cuda)
    beta = (0.2 * torch.randn(c, generator=g)).to(cuda)
    bias = (0.5 * torch.randn(3 * c, generator=g)).to(cuda)
    mask = (torch.rand(n, l - 1, generator=g) < 0.4).half().to(cuda)

    def run(hq):
        out = torch.full((m + 3, c), 7.0, dtype=torch.float16, device=cuda)
        assert lib.oake_debug_set_qkv_walk(hq) == 0
        try:
            if form == 'quad':
                rc = lib.oake_debug_ln_qkv_attn_quad(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     bias.data_ptr(), out.data_ptr(), n, l, heads, _lib.OAKE_F16, None, 1,
                                                     _stream())
            else:
                rc = lib.oake_debug_ln_qkv_attn_obj(x.data_ptr(), w.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                    bias.data_ptr(), mask.data_ptr(), _lib.OAKE_F16, out.data_ptr(), n, l,
                                                    heads, _lib.OAKE_F16, None, 1, _stream())
            assert rc == 0
            torch.cuda.synchronize()
        finally:
            lib.oake_debug_set_qkv_walk(4)
        return out

    base = run(0)
    assert torch.isfinite(base[:m].float()).all() and not torch.equal(base[:m], torch.full_like(base[:m], 7.0))
    assert torch.equal(run(walk), base)


# bit 2: K / V shared through LDS for l > 64; bit 4: persistent loader-wave kernel for l <= 64; bit 5 (63):
# eight-wave blocks for l > 128 (197, 130, 300 below); bit 6 (95): whole K / V in LDS for 64 < l <= 208
# bit 7 (159, the default): one block per head, one-pass softmax for 192 < l <= 208 (197 and the three seam shapes)
@pytest.mark.parametrize('use_tr', [0, 1, 2, 3, 7, 31, 63, 95, 159])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('n,l,heads', [(1, 50, 2), (3, 50, 12), (2, 197, 2), (5, 64, 3), (2, 17, 1),
                                       (1, 130, 1), (3, 77, 8), (2, 65, 1), (1, 300, 2), (1, 1, 1),
                                       (64, 50, 12), (7, 33, 5),
                                       # a last chunk of 49..63 keys: four key tiles, the last one partly valid
                                       (1, 114, 2), (4, 182, 10), (2, 253, 9), (1, 127, 5), (3, 113, 3),
                                       (3, 197, 12), (2, 193, 1), (1, 208, 3), (5, 200, 2), (1, 192, 2), (1, 209, 2)])
def test_attention(lib, lab, cuda, dtype, n, l, heads, use_tr):
    lib = lib if use_tr in (31, 159) else lab
    g = torch.Generator(device='cpu').manual_seed(n * 100 + l + heads)
    qkv = torch.randn(n * l, 3 * heads * 64, generator=g)
    qkv[:, :heads * 64] *= 0.35  # pre-scaled q: scores ~ N(0, 2.8^2): a peaky softmax
    qkv = qkv.to(dtype).to(cuda)
    out = torch.zeros(n * l, heads * 64, dtype=dtype, device=cuda)
    lib.oake_debug_set_attention_variant(use_tr)
    try:
        rc = lib.oake_debug_attention(qkv.data_ptr(), out.data_ptr(), n, l, heads, DT[dtype], _stream())
        assert rc == 0
        torch.cuda.synchronize()
    finally:
        lib.oake_debug_set_attention_variant(159)
    ref = _attention_ref(qkv, n, l, heads)
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    torch.testing.assert_close(out.float(), ref, rtol=tol, atol=tol)


MASK_DT = {torch.float32: _lib.OAKE_F32, torch.float16: _lib.OAKE_F16}


@pytest.mark.parametrize('variant', [31, 159])
@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('mask_dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('n,l,heads', [(2, 197, 2), (5, 197, 12), (1, 194, 1), (3, 201, 3)])
def test_attention_with_object_token(lib, cuda, dtype, mask_dtype, n, l, heads, variant):
    """The objects-mode object token fused into the patch stream's attention (oadp/oake/objects.py:232-247): one
    extra query per sequence over the rows 1..l-1 (bias -100 * mask) and its own key / value.  Both forms the product
    carries: the cooperative kernel's idle wave (31) and attention_head_kernel's own query tile (159)."""
    c = heads * 64
    g = torch.Generator(device='cpu').manual_seed(n * 1000 + l + heads)
    qkv = torch.randn(n * l, 3 * c, generator=g)
    qkv[:, :c] *= 0.35
    qkv_y = torch.randn(n, 3 * c, generator=g)
    qkv_y[:, :c] *= 0.35
    mask = (torch.rand(n, l - 1, generator=g) < 0.4).float()
    mask[0] = 0  # an all-foreground crop
    qkv, qkv_y = qkv.to(dtype).to(cuda), qkv_y.to(dtype).to(cuda)
    mask_d = mask.to(mask_dtype).to(cuda)
    out = torch.zeros(n * l, c, dtype=dtype, device=cuda)
    out_y = torch.zeros(n, c, dtype=dtype, device=cuda)
    lib.oake_debug_set_attention_variant(variant)
    try:
        rc = lib.oake_debug_attention_objects(qkv.data_ptr(), qkv_y.data_ptr(), mask_d.data_ptr(), MASK_DT[mask_dtype],
                                              out.data_ptr(), out_y.data_ptr(), n, l, heads, DT[dtype], _stream())
        if variant == 31 and l > 128 + 96:
            assert rc == _lib.OAKE_ERR_UNSUPPORTED
            return
        assert rc == 0
        torch.cuda.synchronize()
    finally:
        lib.oake_debug_set_attention_variant(159)
    tol = 3e-3 if dtype == torch.float16 else 2e-2
    torch.testing.assert_close(out.float(), _attention_ref(qkv, n, l, heads), rtol=tol, atol=tol)
    q, k, v = qkv.float().view(n, l, 3, heads, 64).permute(2, 0, 3, 1, 4)          # [n, h, l, 64]
    qy, ky, vy = qkv_y.float().view(n, 1, 3, heads, 64).permute(2, 0, 3, 1, 4)     # [n, h, 1, 64]
    keys, vals = torch.cat([k[:, :, 1:], ky], 2), torch.cat([v[:, :, 1:], vy], 2)
    bias = torch.cat([-100.0 * mask.to(cuda), torch.zeros(n, 1, device=cuda)], 1)[:, None, None, :]
    p = torch.softmax(qy @ keys.transpose(-1, -2) + bias, dim=-1)
    ref_y = (p @ vals).permute(0, 2, 1, 3).reshape(n, c)
    torch.testing.assert_close(out_y.float(), ref_y, rtol=tol, atol=tol)


def test_tr_read_semantics(lib, cuda):
    """ds_read_b64_tr_b16: lane i of a 16-lane group, addressing the 8-B slice i of a row-major
    4x16 block of 16-bit elements, receives column i of that block."""
    src = torch.arange(256, dtype=torch.int16, device=cuda)
    out = torch.zeros(256, dtype=torch.int16, device=cuda)
    assert lib.oake_debug_tr_read(src.data_ptr(), out.data_ptr(), _stream()) == 0
    torch.cuda.synchronize()
    got = out.view(64, 4).cpu()
    lanes = torch.arange(64)
    exp = torch.stack([64 * (lanes // 16) + 16 * j + (lanes % 16) for j in range(4)], dim=1)
    assert torch.equal(got, exp.to(torch.int16)), f'tr-read mapping differs:\n{got}'


def test_mfma_probe_counts_its_work(lib, cuda):
    """oake_debug_mfma_probe (bench.py's power-cap probe): runs, reports CUs x 8 waves x 20 MFMAs x 16384 FLOP
    per iteration, and leaves its sink alone."""
    import ctypes as C
    frags = (torch.randn(9 * 64 * 8) * 0.5).half().to(cuda)
    sink = torch.zeros(1, device=cuda)
    flop = C.c_double(0)
    assert lib.oake_debug_mfma_probe(frags.data_ptr(), sink.data_ptr(), 100, C.byref(flop), _stream()) == 0
    torch.cuda.synchronize()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert flop.value == cus * 8 * 20 * 16384.0 * 100 and sink.item() == 0.0
    assert lib.oake_debug_mfma_probe(0, sink.data_ptr(), 100, None, _stream()) != 0


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('n,l', [(1, 50), (3, 50), (300, 50), (2, 64), (5, 49), (4, 48), (2, 33), (3, 17), (2, 16),
                                 (3, 1), (257, 50)])
def test_attn_out_fused(lab, cuda, dtype, n, l):
    lib = lab  # (the kernel lost its A/B: lab library only)
    """csrc/attn_out.hip: attention + out_proj + bias + residual (16-bit, in place) + the row statistics of the next
    LayerNorm in one kernel, against the same chain in fp32 torch from the same 16-bit operands.  The attention
    output is rounded to 16 bits before out_proj in both (the kernel keeps it as MFMA operand fragments)."""
    heads, c = 12, 768
    g = torch.Generator(device='cpu').manual_seed(n * 100 + l)
    qkv = torch.randn(n * l, 3 * c, generator=g)
    qkv[:, :c] *= 0.35
    qkv = qkv.to(dtype).to(cuda)
    w = (torch.randn(c, c, generator=g) * c ** -0.5).to(dtype).to(cuda)
    bias = torch.randn(c, generator=g).to(cuda)
    x0 = torch.randn(n * l, c, generator=g).to(dtype).to(cuda)
    x = torch.cat([x0, torch.full((3, c), 7.0, dtype=dtype, device=cuda)])  # guard rows behind the last image
    part = torch.full((n * l + 3, 16, 2), float('nan'), device=cuda)
    rc = lib.oake_debug_attn_out(qkv.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), part.data_ptr(),
                                 n, l, heads, DT[dtype], _stream())
    assert rc == 0
    torch.cuda.synchronize()
    att = _attention_ref(qkv, n, l, heads).to(dtype).float()
    ref = x0.float() + att @ w.float().t() + bias
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    torch.testing.assert_close(x[:n * l].float(), ref, rtol=tol, atol=tol)
    assert torch.equal(x[n * l:], torch.full((3, c), 7.0, dtype=dtype, device=cuda))  # nothing written past the rows
    # (sum, sum of squares) of the fp32 values before rounding, per 64-column slice; slots 12..15 untouched
    sl = ref.view(n * l, 12, 64)
    torch.testing.assert_close(part[:n * l, :12, 0], sl.sum(-1), rtol=2e-2, atol=0.08)
    torch.testing.assert_close(part[:n * l, :12, 1], (sl * sl).sum(-1), rtol=2e-2, atol=0.3)
    assert torch.isnan(part[:n * l, 12:]).all() and torch.isnan(part[n * l:]).all()


def test_attn_out_refuses_other_geometries(lib, lab, cuda):
    z = torch.zeros(64, device=cuda)
    # the product library does not carry the kernel at all
    assert lib.oake_debug_attn_out(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 50, 12,
                                   _lib.OAKE_F16, _stream()) == _lib.OAKE_ERR_UNSUPPORTED
    lib = lab
    assert lib.oake_debug_attn_out(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 65, 12,
                                   _lib.OAKE_F16, _stream()) == _lib.OAKE_ERR_UNSUPPORTED
    assert lib.oake_debug_attn_out(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 50, 8,
                                   _lib.OAKE_F16, _stream()) == _lib.OAKE_ERR_UNSUPPORTED


def test_production_library_refuses_lab_variants(lib, lab):
    """the product carries only what its own selection can return; the experiments are in the
    lab build, and both say which they are."""
    assert lib.oake_debug_lab_build() == 0 and lab.oake_debug_lab_build() == 1
    for v in range(-2, 14):
        want = _lib.OAKE_OK if v in PROD_GEMM else _lib.OAKE_ERR_UNSUPPORTED
        assert lib.oake_debug_set_gemm_variant(v) == want, v
        assert lab.oake_debug_set_gemm_variant(v) == _lib.OAKE_OK
    assert lib.oake_debug_set_gemm_variant(14) == lab.oake_debug_set_gemm_variant(14) == _lib.OAKE_ERR_UNSUPPORTED
    lib.oake_debug_set_gemm_variant(-1)
    lab.oake_debug_set_gemm_variant(-1)
    for v in (0, 7, 30, 63, 95, 128, 191, 255, 256, -1):
        assert lib.oake_debug_set_attention_variant(v) == _lib.OAKE_ERR_UNSUPPORTED
    assert lib.oake_debug_set_attention_variant(31) == _lib.OAKE_OK  # (the cooperative kernel at 197 keys: A/B runs)
    assert lib.oake_debug_set_attention_variant(159) == _lib.OAKE_OK
    assert lab.oake_debug_set_attention_variant(95) == _lib.OAKE_OK and lab.oake_debug_set_attention_variant(255) == _lib.OAKE_OK
    assert lab.oake_debug_set_attention_variant(159) == _lib.OAKE_OK
