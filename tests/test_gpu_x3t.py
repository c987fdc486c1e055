"""bf16x3 by operand expansion (round 4; include/sdn.h): the tolerance-meeting mode's contractions on the LDS-DMA tiles of the
16-bit engine.  A tensor a GEMM reads is the bf16 triple [hi | lo | hi], its weight [hi | hi | lo] per K-group, and
A'.W'^T = a_hi w_hi + a_lo w_hi + a_hi w_lo is one bf16 GEMM with three times the k loop.  Operator level here (vs float64
torch on the CPU; per-operator bound 3e-5 as for sdn_gemm_x3, measured ~1e-5); the network-level bounds are those of the
bf16x3 plan in tests/test_gpu_f32.py and tests/test_gpu_e2e_ids.py, which now run on this path."""
import pytest
import torch
import torch.nn.functional as F

import safe_denoiser_amd as sda
from safe_denoiser_amd import _lib
from safe_denoiser_amd.unet import _interleave16
from tests_support import ops

pytestmark = pytest.mark.gpu
TOL = 3e-5


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_split3_and_expand3_layout_and_precision():
    x, x2 = rnd(37, 64, seed=1) * 3.0, rnd(37, 128, seed=2) * 1e-3
    t = ops.split3(x.cuda(), x2.cuda())
    assert t.shape == (37, 3 * 192)
    v = ops.triple_value(t).cpu()
    ref = torch.cat([x, x2], 1).double()
    assert float(((v - ref).abs() / ref.abs().clamp_min(1e-30)).max()) <= 2.0 ** -16        # hi + lo carries 16 mantissa bits
    assert torch.equal(t[:, :192].cpu(), ref.float().bfloat16())                             # hi = round-to-nearest-even bf16
    w = rnd(10, 9 * 64, seed=3)
    e = ops.expand3(w.cuda(), group=64).cpu().reshape(10, 9, 3, 64)                          # conv weight: per-tap [hi | hi | lo]
    hi = w.bfloat16().reshape(10, 9, 64)
    assert torch.equal(e[:, :, 0], hi) and torch.equal(e[:, :, 1], hi)
    assert torch.equal(e[:, :, 2], (w.reshape(10, 9, 64) - hi.float()).bfloat16())
    e2 = ops.expand3(w.cuda()).cpu()                                                         # plain matrix: group = K
    assert torch.equal(e2[:, :576], w.bfloat16()) and torch.equal(e2[:, 576:1152], w.bfloat16())


@pytest.mark.parametrize("M,N,K", [(128, 320, 320), (4113, 320, 320), (1000, 960, 1280), (3000, 640, 1280), (2, 1280, 320)])
def test_x3t_gemm_f32_out_with_bias_and_f32_residual(M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a.double() @ w.double().T + bias.double()
    a3, w3 = ops.split3(a.cuda()), ops.expand3(w.cuda())
    assert rel_l2(ops.gemm_x3t(a3, w3, N, K, bias=bias.cuda()), ref) <= TOL
    out = ops.gemm_x3t(a3, w3, N, K, bias=bias.cuda(), residual=res.cuda())
    assert rel_l2(out, ref + res.double()) <= TOL
    tri = ops.gemm_x3t(a3, w3, N, K, bias=bias.cuda(), residual=res.cuda(), x3_out=3)       # the next GEMM's operand directly
    assert rel_l2(ops.triple_value(tri), ref + res.double()) <= TOL
    assert torch.equal(tri[:, :N], out.bfloat16())                                          # = the split of the f32 result


def test_x3t_gemm_two_source_concat_and_geglu_triple_out():
    M, K1, K2, N = 700, 640, 320, 320
    a1, a2 = rnd(M, K1, seed=5), rnd(M, K2, seed=6)
    w = rnd(N, K1 + K2, seed=7, scale=(K1 + K2) ** -0.5)
    ref = torch.cat([a1, a2], 1).double() @ w.double().T
    out = ops.gemm_x3t(ops.split3(a1.cuda(), a2.cuda()), ops.expand3(w.cuda()), N, K1 + K2)
    assert rel_l2(out, ref) <= TOL
    C_ = 320
    x, w1, b1 = rnd(M, C_, seed=12), rnd(8 * C_, C_, seed=13, scale=C_ ** -0.5), rnd(8 * C_, seed=14)
    val, gate = (x.double() @ w1.double().T + b1.double()).chunk(2, -1)
    tri = ops.gemm_x3t(ops.split3(x.cuda()), ops.expand3(_interleave16(w1).contiguous().cuda()), 8 * C_, C_,
                       bias=_interleave16(b1).contiguous().cuda(), act=2, x3_out=2)
    assert tri.shape == (M, 3 * 4 * C_)
    assert rel_l2(ops.triple_value(tri), val * F.gelu(gate)) <= TOL


@pytest.mark.parametrize("B,H,Cin,Cout,stride,ups,asym", [(2, 16, 320, 320, 1, 0, 0), (1, 16, 640, 320, 2, 0, 0), (2, 8, 320, 640, 1, 1, 0),
                                                          (48, 32, 320, 320, 1, 0, 0),      # M = 49152: the slab-ring kernel
                                                          (2, 16, 128, 128, 2, 0, 1)])
def test_x3t_conv3x3_with_rowbias_and_residual(B, H, Cin, Cout, stride, ups, asym):
    x, w, bias = rnd(B, Cin, H, H, seed=15), rnd(Cout, Cin, 3, 3, seed=16, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=17)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = (F.conv2d(F.pad(xin, (0, 1, 0, 1)).double(), w.double(), bias.double(), stride=2) if asym
           else F.conv2d(xin.double(), w.double(), bias.double(), stride=stride, padding=1))
    Ho = ref.shape[-1]
    rb = rnd(B, Cout, seed=18)
    res = rnd(B * Ho * Ho, Cout, seed=19)
    ref = ref + rb.double()[:, :, None, None] + res.double().reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    xn = x.permute(0, 2, 3, 1).reshape(B * H * H, Cin).contiguous().cuda()
    a3 = ops.split3(xn).reshape(B, H * H, 3 * Cin)
    w3 = ops.expand3(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda(), group=Cin)
    sda.lib().sdn_debug_gemm_launch_counts(None, 1)
    out = ops.gemm_x3t(a3, w3, Cout, 9 * Cin, bias=bias.cuda(), rowbias=rb.cuda(), residual=res.cuda(),
                       conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=Ho, Wo=Ho, stride=stride, upsample=ups, asym_pad=asym))
    assert rel_l2(out.reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2), ref) <= TOL
    if B == 48:                                                                             # not vacuous: the slab kernel took it
        import ctypes as C
        cnt = (C.c_longlong * 2)()
        sda.lib().sdn_debug_gemm_launch_counts(cnt, 0)
        assert cnt[0] == 1 and cnt[1] == 0


def test_x3t_conv_out_nchw_and_argument_checks():
    B, H, Cin = 2, 16, 320
    x, w, bias = rnd(B, Cin, H, H, seed=25), rnd(4, Cin, 3, 3, seed=26, scale=(9 * Cin) ** -0.5), rnd(4, seed=27)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    wp = torch.zeros(32, 9 * Cin); wp[:4] = w.permute(0, 2, 3, 1).reshape(4, 9 * Cin)
    bp = torch.zeros(32); bp[:4] = bias
    a3 = ops.split3(x.permute(0, 2, 3, 1).reshape(B * H * H, Cin).contiguous().cuda()).reshape(B, H * H, 3 * Cin)
    out = ops.gemm_x3t(a3, ops.expand3(wp.cuda(), group=Cin), 32, 9 * Cin, bias=bp.cuda(), x3_out=0, out_kind=2, n_valid=4,
                       conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H))
    assert rel_l2(out.reshape(B, 4, H, H), ref) <= TOL
    a3, w3 = ops.split3(rnd(64, 64).cuda()), ops.expand3(rnd(64, 64).cuda())
    with pytest.raises(_lib.SdnError):
        ops.gemm_x3t(a3, w3, 64, 64, x3_out=2)                                              # triple GEGLU output without the GEGLU act
    with pytest.raises(_lib.SdnError):
        ops.gemm_x3t(a3, w3, 64, 64, act=1)                                                 # no SiLU epilogue on this path


def test_norms_and_attention_write_the_triple_of_their_f32_result():
    L = sda.lib()
    B, hw, c1, c2 = 2, 256, 320, 640
    x, x2 = (rnd(B, hw, c1, seed=20) + 3.0).cuda(), (rnd(B, hw, c2, seed=21, scale=0.1) + 50.0).cuda()
    g, b = rnd(c1 + c2, seed=22).cuda(), rnd(c1 + c2, seed=23).cuda()
    f = ops.groupnorm(x, x2, 32, 1e-5, 1, g, b)
    t = torch.empty((B, hw, 3 * (c1 + c2)), dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(B * 129 * 32 * 2, dtype=torch.float32, device="cuda")
    _lib.check(L.sdn_groupnorm_f32_triple(x.data_ptr(), x2.data_ptr(), B, hw, c1, c2, 32, 1e-5, 1, g.data_ptr(), b.data_ptr(), t.data_ptr(),
                                          ws.data_ptr(), _lib.stream_ptr()), "gn triple")
    assert torch.equal(t.reshape(B * hw, -1), ops.split3(f.reshape(B * hw, c1 + c2)))
    y, gl, bl = (rnd(1001, 1280, seed=24) + 2.0).cuda(), rnd(1280, seed=25).cuda(), rnd(1280, seed=26).cuda()
    t = torch.empty((1001, 3 * 1280), dtype=torch.bfloat16, device="cuda")
    _lib.check(L.sdn_layernorm_f32_triple(y.data_ptr(), 1001, 1280, 1e-5, gl.data_ptr(), bl.data_ptr(), t.data_ptr(), _lib.stream_ptr()), "ln triple")
    assert torch.equal(t, ops.split3(ops.layernorm(y, gl, bl)))
    for nq, nk, d in ((256, 256, 40), (100, 77, 80), (64, 77, 160)):
        H = 8
        q, kv = rnd(B, nq, H * d, seed=30).cuda(), rnd(B, nk, 2 * H * d, seed=31).cuda()
        ops.X3 = True
        try:
            f = ops.attention(q, kv[..., :H * d], kv[..., H * d:], H)
        finally:
            ops.X3 = False
        t = torch.empty((B, nq, 3 * H * d), dtype=torch.bfloat16, device="cuda")
        _lib.check(L.sdn_attention_x3_triple(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * H * d, t.data_ptr(), B, H, nq, nk, d, H * d,
                                             2 * H * d, 2 * H * d, H * d, d ** -0.5, _lib.stream_ptr()), "attention triple")
        assert torch.equal(t.reshape(B * nq, -1), ops.split3(f.reshape(B * nq, H * d))), (nq, nk, d)


def test_x3t_gemm_pair_rows_are_the_split_of_the_f32_result():
    M, N, K = 1000, 960, 320
    a, w, bias = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43)
    a3, w3 = ops.split3(a.cuda()), ops.expand3(w.cuda())
    f = ops.gemm_x3t(a3, w3, N, K, bias=bias.cuda())
    pr = ops.gemm_x3t(a3, w3, N, K, bias=bias.cuda(), x3_out=4)
    assert pr.shape == (M, 2 * N)
    hi = f.bfloat16()
    assert torch.equal(pr[:, :N], hi)
    assert torch.equal(pr[:, N:], (f - hi.float()).bfloat16())


@pytest.mark.parametrize("B,N,d", [(2, 256, 40), (1, 300, 40), (1, 4096, 40), (2, 1024, 80), (1, 100, 80), (8, 128, 40),
                                   (2, 256, 160), (1, 64, 160), (1, 300, 160)])
def test_attention_on_presplit_pairs_against_float64(B, N, d):
    """sdn_attention_x3_pairs (K / V planes by LDS-DMA, three bf16 products per term) vs float64 softmax attention on the values the
    pairs carry, and vs the first bf16x3 kernel (sdn_attention_x3 on the f32 tensor); triple output = split of the f32 output."""
    H = 8
    C_ = H * d
    qkv = rnd(B, N, 3 * C_, seed=44)
    qkv[0, 5, :d] *= 4.0                                                         # a sharp row (running-maximum rescale after tile 0)
    qkv[0, min(N - 1, 200), C_:C_ + d] = qkv[0, 5, :d] * 1.5
    g = qkv.cuda()
    hi = g.bfloat16()
    pairs = torch.cat([hi, (g - hi.float()).bfloat16()], -1).contiguous()
    val = (pairs[..., :3 * C_].double() + pairs[..., 3 * C_:].double()).cpu()    # what the kernel is given
    sp = lambda t: t.reshape(B, N, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(val[..., :C_]), sp(val[..., C_:2 * C_]), sp(val[..., 2 * C_:])).transpose(1, 2).reshape(B, N, C_)
    out = ops.attention_x3_pairs(pairs, H)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) <= TOL, rel_l2(out, ref)
    ops.X3 = True
    try:
        old = ops.attention(g[..., :C_], g[..., C_:2 * C_], g[..., 2 * C_:], H)
    finally:
        ops.X3 = False
    assert rel_l2(out, old) <= 2 * TOL
    tri = ops.attention_x3_pairs(pairs, H, triple=True)
    assert torch.equal(tri.reshape(B * N, -1), ops.split3(out.reshape(B * N, C_)))


@pytest.mark.parametrize("B,Nq,Nk,d", [(2, 256, 77, 40), (1, 4096, 77, 40), (2, 1024, 77, 80), (1, 100, 77, 160), (8, 256, 77, 160), (1, 64, 200, 40)])
def test_cross_attention_on_presplit_pairs_against_float64(B, Nq, Nk, d):
    """Cross-attention form of sdn_attention_x3_pairs: queries from a [hi | lo] projection of width C, keys / values from the text
    projection's pair rows [hi(k | v) | lo(k | v)] (different row strides and lo offsets); ragged key tail (77 keys)."""
    H = 8
    C_ = H * d
    q, kv = rnd(B, Nq, C_, seed=51).cuda(), rnd(B, Nk, 2 * C_, seed=52).cuda()
    pair = lambda t: torch.cat([t.bfloat16(), (t - t.bfloat16().float()).bfloat16()], -1).contiguous()
    qp, kvp = pair(q), pair(kv)
    qv = (qp[..., :C_].double() + qp[..., C_:].double()).cpu()
    kvv = (kvp[..., :2 * C_].double() + kvp[..., 2 * C_:].double()).cpu()
    sp = lambda t, n: t.reshape(B, n, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(qv, Nq), sp(kvv[..., :C_], Nk), sp(kvv[..., C_:], Nk)).transpose(1, 2).reshape(B, Nq, C_)
    out = ops.cross_attention_x3_pairs(qp, kvp, H)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) <= TOL, rel_l2(out, ref)


@pytest.mark.parametrize("M,N,K", [(49152, 1280, 1280), (24576, 2560, 640), (61440, 640, 512)])
def test_experimental_h8_operand_form_against_float64(M, N, K):
    """sdn_gemm_desc.x3_out = 5 (EXPERIMENTAL): rows [fp16(a) | e4m3(2^11 lo(a)) | e4m3(a)] against
    [fp16(w) | e4m3(w) | e4m3(2^11 lo(w))] -- an fp16 main term plus two correction products on v_mfma_scale_f32_16x16x128_f8f6f4 with
    the 2^-11 in the instruction's block scale, f32 accumulation into one set of accumulators.  Against float64 on activations with
    outlier channels: <= 4e-5 (measured 1.2e-5 ... 2.2e-5; the fp16 GEMM alone: 2.9e-4), i.e. the corrections really land, with the
    right scale, in both tile widths (N % 256 == 0 -> 256-wide, else 320-wide); what the form does not cover is refused."""
    import ctypes as C
    import safe_denoiser_amd as sda
    from safe_denoiser_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn(M, K, device="cuda", generator=g)
    a[:, ::64] *= 12.0
    w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    ah, wh = a.half(), w.half()
    q = lambda x: x.to(torch.float8_e4m3fn).view(torch.uint8)
    a8 = torch.cat([ah.view(torch.uint8), q((a - ah.float()) * 2048.0), q(ah.float())], dim=1).contiguous()
    w8 = torch.cat([wh.view(torch.uint8), q(wh.float()), q((w - wh.float()) * 2048.0)], dim=1).contiguous()
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.out_kind, d.x3_out, d.ldc = M, N, 2 * K, 1, 5, N
    call = lambda fn, dd: fn(C.byref(dd), a8.data_ptr(), None, w8.data_ptr(), None, None, None, None, out.data_ptr(), _lib.stream_ptr())
    assert call(sda.lib().sdn_gemm_f16, d) == 0
    rows = torch.arange(0, M, M // 1024, device="cuda")[:1024]
    ref = a[rows].double() @ w.double().T
    r8 = float((out[rows].double() - ref).norm() / ref.norm())
    r16 = float(((ah[rows].float() @ wh.float().T).double() - ref).norm() / ref.norm())
    print(f"h8 GEMM {M} x {N} x {K}: rel L2 vs float64 {r8:.2e} (fp16 operands alone {r16:.2e})")
    assert r8 <= 4e-5 and r16 > 5 * r8
    assert call(sda.lib().sdn_gemm_bf16, d) != 0                          # the fp16 instance only
    d.act = 1
    assert call(sda.lib().sdn_gemm_f16, d) != 0                           # no activation
    d.act, d.K = 0, 2 * K + 64
    assert call(sda.lib().sdn_gemm_f16, d) != 0                           # logical K % 128 == 0
