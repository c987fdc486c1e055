"""Operator-level parity: each HIP kernel (through the C ABI) vs a plain PyTorch fp32 reference of the same op on the
SAME bf16-rounded inputs.  Expected error = one bf16 rounding of the output (2^-9 relative) + fp32 accumulation-order
noise; tolerances are written per test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import safe_denoiser_amd as sda
from safe_denoiser_amd import _lib
from tests_support import ops

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def check_bf16(out, ref, tol=4e-3):
    """bf16 output vs fp32 reference: relative L2 <= tol and max elementwise error <= 2^-7 of the reference scale."""
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert rel_l2(out, ref) <= tol, rel_l2(out, ref)
    err = (out.float().cpu() - ref.float().cpu()).abs().max()
    assert float(err) <= 2 ** -7 * float(ref.abs().max()) + 1e-3, float(err)


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 320, 320), (256, 160, 64), (4096, 640, 1280), (77 * 3, 1280, 768),
                                   (2, 1280, 320), (192, 128, 128), (4096 * 2, 960, 320), (64, 64, 64), (100, 32, 64)])
def test_gemm_plain_bias(M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    ref = a.float() @ w.float().T + bias
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda())
    check_bf16(out, ref)


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric W catches a transposed C write (guide: 'A=I-check with ASYMMETRIC B')."""
    K = 128
    a = torch.eye(K).to(BF)
    w = (torch.arange(160 * K).reshape(160, K) % 251 - 125).float().to(BF)
    out = ops.gemm(a.cuda(), w.cuda())
    assert torch.equal(out.float().cpu(), w.float().T.contiguous())


def test_gemm_f32_out_residual_silu_and_dual_source():
    M, N, K1, K2 = 384, 320, 640, 320
    a1, a2 = rnd(M, K1, seed=4), rnd(M, K2, seed=5)
    w = rnd(N, K1 + K2, seed=6, scale=(K1 + K2) ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(7))
    res = rnd(M, N, seed=8)
    ref = torch.cat([a1, a2], 1).float() @ w.float().T + bias
    out = ops.gemm(a1.cuda(), w.cuda(), bias=bias.cuda(), a2=a2.cuda())                 # skip-concat, never materialised
    check_bf16(out, ref)
    out = ops.gemm(a1.cuda(), w.cuda(), bias=bias.cuda(), a2=a2.cuda(), residual=res.cuda())
    check_bf16(out, ref + res.float())
    out = ops.gemm(a1.cuda(), w.cuda(), bias=bias.cuda(), a2=a2.cuda(), act=1)
    check_bf16(out, F.silu(ref))
    out = ops.gemm(a1.cuda(), w.cuda(), bias=bias.cuda(), a2=a2.cuda(), out_kind=1)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-4, atol=2e-4)                    # f32 out: accumulation order only


def test_gemm_rowbias_per_sample():
    B, hw, N, K = 3, 64, 320, 320
    a, w = rnd(B * hw, K, seed=9), rnd(N, K, seed=10, scale=K ** -0.5)
    rb = torch.randn(B, 2 * N, generator=torch.Generator().manual_seed(11))
    ref = (a.float() @ w.float().T).reshape(B, hw, N) + rb[:, None, N:]
    rbg = rb.cuda()
    out = ops.gemm(a.cuda(), w.cuda(), rowbias=rbg[:, N:], rows_per_batch=hw)            # a column slice of the stacked time proj
    check_bf16(out.reshape(B, hw, N), ref)


@pytest.mark.parametrize("C", [320, 640])
def test_gemm_geglu(C):
    from safe_denoiser_amd.unet import _interleave16
    M = 256
    x = rnd(M, C, seed=12)
    w = rnd(8 * C, C, seed=13, scale=C ** -0.5)
    b = torch.randn(8 * C, generator=torch.Generator().manual_seed(14))
    proj = x.float() @ w.float().T + b
    val, gate = proj.chunk(2, -1)
    ref = val * F.gelu(gate)
    out = ops.gemm(x.cuda(), _interleave16(w).contiguous().cuda(), bias=_interleave16(b).contiguous().cuda(), act=2)
    check_bf16(out, ref)


@pytest.mark.parametrize("B,H,Cin,Cout,stride,ups", [(2, 16, 320, 320, 1, 0), (1, 16, 640, 320, 2, 0),
                                                     (2, 8, 320, 640, 1, 1), (3, 8, 64, 160, 1, 0),
                                                     (1, 64, 320, 320, 1, 0), (5, 4, 64, 64, 1, 0), (3, 8, 64, 64, 2, 0),
                                                     (2, 4, 64, 96, 1, 1)])   # output maps narrower than 8: per-piece decode
def test_conv3x3_implicit_gemm(B, H, Cin, Cout, stride, ups):
    x = rnd(B, Cin, H, H, seed=15)
    w = rnd(Cout, Cin, 3, 3, seed=16, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(17))
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if ups else x.float()
    ref = F.conv2d(xin, w.float(), bias, stride=stride, padding=1)
    Ho = ref.shape[-1]
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()                                       # NHWC
    wn = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()                # [O][ky][kx][I]
    out = ops.gemm(xn, wn, bias=bias.cuda(), conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=Ho, Wo=Ho, stride=stride, upsample=ups))
    check_bf16(out.reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("B,H,Cin,Cout", [(2, 16, 128, 128), (1, 32, 64, 256)])
def test_conv3x3_stride2_with_right_bottom_padding(B, H, Cin, Cout):
    """The VAE encoder's Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then conv3x3 stride 2 (diffusers-0.29.0)."""
    x = rnd(B, Cin, H, H, seed=41)
    w = rnd(Cout, Cin, 3, 3, seed=42, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(43))
    ref = F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), bias, stride=2, padding=0)
    Ho = ref.shape[-1]
    assert Ho == H // 2
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wn = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    out = ops.gemm(xn, wn, bias=bias.cuda(), conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=Ho, Wo=Ho, stride=2, asym_pad=1))
    check_bf16(out.reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2), ref)
    with pytest.raises(sda.SdnError):                                # asym_pad is a stride-2 mode only
        ops.gemm(xn, wn, bias=bias.cuda(), conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H, stride=1, asym_pad=1))


@pytest.mark.parametrize("split", [2, 5, 16])
def test_split_k_matches_unsplit_gemm_and_conv(split):
    """sdn_gemm_splitk_*: the k loop cut into slices + a deterministic reduce pass gives the unsplit result up to fp32
    summation order (checked against the unsplit kernel at 2e-3 and against torch at the usual one-rounding bound)."""
    # plain GEMM with bias, per-sample row bias, residual and SiLU; M tail not a multiple of the tile
    M, N, K = 200, 320, 2048
    a, w = rnd(M, K, seed=51), rnd(N, K, seed=52, scale=K ** -0.5)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(53))
    rb = torch.randn(2, N, generator=torch.Generator().manual_seed(54))
    res = rnd(M, N, seed=55)
    args = dict(bias=bias.cuda(), rowbias=rb.cuda(), residual=res.cuda(), rows_per_batch=100, act=1)
    o_ref = ops.gemm(a.cuda(), w.cuda(), **args)
    o_spl = ops.gemm(a.cuda(), w.cuda(), split_k=split, **args)
    assert rel_l2(o_spl, o_ref) <= 2e-3
    lin = a.float() @ w.float().t() + bias + rb.repeat_interleave(100, 0)
    check_bf16(o_spl, F.silu(lin + res.float()))
    # 3x3 conv (k order = channel chunk outer, tap inner: a slice may start mid-chunk)
    B, H, Cin, Cout = 2, 8, 320, 320
    x = rnd(B, Cin, H, H, seed=56); wc = rnd(Cout, Cin, 3, 3, seed=57, scale=(9 * Cin) ** -0.5)
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); wn = wc.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    cv = dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H)
    o_spl = ops.gemm(xn, wn, bias=bias.cuda(), conv=cv, split_k=split)
    check_bf16(o_spl.reshape(B, H, H, Cout).permute(0, 3, 1, 2), F.conv2d(x.float(), wc.float(), bias, padding=1))
    # two-source A (skip concat) with the seam inside a slice
    a1, a2 = rnd(M, 640, seed=58), rnd(M, 1280, seed=59)
    w2 = rnd(N, 1920, seed=60, scale=1920 ** -0.5)
    o_spl = ops.gemm(a1.cuda(), w2.cuda(), a2=a2.cuda(), split_k=split)
    check_bf16(o_spl, torch.cat([a1, a2], 1).float() @ w2.float().t())


def test_split_k_rejects_what_it_cannot_do():
    a, w = rnd(64, 1024).cuda(), rnd(64, 1024).cuda()
    with pytest.raises(sda.SdnError):
        ops.gemm(a, w, split_k=32)                                   # more slices than k-tiles
    with pytest.raises(sda.SdnError):
        ops.gemm(a, w, split_k=2, out_kind=1)                        # fp32 output
    d = _lib.GemmDesc(); d.M, d.N, d.K, d.split_k = 64, 64, 1024, 2
    o = torch.empty(64, 64, dtype=BF, device="cuda")
    assert sda.lib().sdn_gemm_bf16(C.byref(d), a.data_ptr(), None, w.data_ptr(), None, None, None, None, o.data_ptr(), _lib.stream_ptr()) != 0


@pytest.mark.parametrize("M,N,K,geglu", [(300, 320, 320, False), (1000, 960, 640, False), (4096, 3840, 1280, False),
                                         (520, 2560, 320, True), (2048, 5120, 640, True), (64, 640, 640, False)])
def test_layernorm_folded_into_gemm(M, N, K, geglu):
    """sdn_ln_fold + sdn_gemm_ln_*: LayerNorm never materialised.  Reference = torch layer_norm (fp32) then the linear
    with the 16-bit weights; rows get a large common offset to exercise the mean cancellation."""
    g = torch.Generator().manual_seed(70)
    x = (rnd(M, K, seed=71) * 1.5 + torch.randn(M, 1, generator=g) * 2.0).to(BF)
    w = rnd(N, K, seed=72, scale=K ** -0.5)
    gamma = 1 + 0.2 * torch.randn(K, generator=g); beta = 0.3 * torch.randn(K, generator=g)
    bias = torch.randn(N, generator=g)
    y = F.linear(F.layer_norm(x.float(), (K,), gamma, beta, 1e-5), w.float(), bias)
    if geglu:
        from safe_denoiser_amd.unet import _interleave16
        ref = y[:, :N // 2] * F.gelu(y[:, N // 2:])
        out = ops.gemm_ln(x.cuda(), _interleave16(w).contiguous().cuda(), gamma.cuda(), beta.cuda(),
                          _interleave16(bias).contiguous().cuda(), act=2, prepass=True)
    else:
        ref = y
        out = ops.gemm_ln(x.cuda(), w.cuda(), gamma.cuda(), beta.cuda(), bias.cuda(), prepass=N > 640)
        check_bf16(ops.gemm_ln(x.cuda(), w.cuda(), gamma.cuda(), beta.cuda(), bias.cuda(), prepass=True), ref, tol=6e-3)
    check_bf16(out, ref, tol=6e-3)
    if not geglu:                                                    # no bias: d = sum_k beta W only
        out0 = ops.gemm_ln(x.cuda(), w.cuda(), gamma.cuda(), beta.cuda(), None, prepass=N > 640)
        check_bf16(out0, y - bias, tol=6e-3)


def test_groupnorm_statistics_from_the_producing_gemms():
    """sdn_gemm_stats_* + sdn_groupnorm_cols_*: the GEMMs that write a GroupNorm's inputs also leave per-128-row-block column
    sums, and the GroupNorm reduces those instead of reading the tensors.  Checked on a channel concat of a conv output
    (256-row tile, two staging passes) and a linear output with residual (128-row tile), against the plain GroupNorm of
    the very same tensors (bit-for-bit inputs) and torch."""
    B, H = 3, 16                                                      # hw = 256 = two 128-row blocks per sample
    hw, M = H * H, B * H * H
    x = rnd(B, 640, H, H, seed=81); wc = rnd(320, 640, 3, 3, seed=82, scale=(9 * 640) ** -0.5)
    bias = torch.randn(320, generator=torch.Generator().manual_seed(83))
    xn = x.permute(0, 2, 3, 1).contiguous().cuda(); wn = wc.permute(0, 2, 3, 1).reshape(320, 9 * 640).contiguous().cuda()
    cols1 = torch.zeros(M // 128, 320, 2, device="cuda")
    y1 = ops.gemm(xn, wn, bias=bias.cuda(), conv=dict(Hs=H, Ws=H, Cin=640, Ho=H, Wo=H), col_stats=cols1)
    a = rnd(M, 320, seed=84); w2 = rnd(640, 320, seed=85, scale=320 ** -0.5); res = rnd(M, 640, seed=86)
    cols2 = torch.zeros(M // 128, 640, 2, device="cuda")
    y2 = ops.gemm(a.cuda(), w2.cuda(), residual=res.cuda(), col_stats=cols2)
    # the partials are exactly the column sums of the stored values
    ref1 = y1.float().reshape(M // 128, 128, 320)
    torch.testing.assert_close(cols1[..., 0], ref1.sum(1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(cols1[..., 1], (ref1 * ref1).sum(1), rtol=1e-5, atol=1e-3)
    gamma = (1 + 0.1 * torch.randn(960, generator=torch.Generator().manual_seed(87))).cuda()
    beta = (0.1 * torch.randn(960, generator=torch.Generator().manual_seed(88))).cuda()
    g_plain = ops.groupnorm(y1.reshape(B, hw, 320), y2.reshape(B, hw, 640), 32, 1e-5, 1, gamma, beta)
    g_cols = ops.groupnorm(y1.reshape(B, hw, 320), y2.reshape(B, hw, 640), 32, 1e-5, 1, gamma, beta, cols1=cols1, cols2=cols2)
    assert rel_l2(g_cols, g_plain) <= 1e-3
    full = torch.cat([y1.reshape(B, hw, 320), y2.reshape(B, hw, 640)], 2).float().cpu().permute(0, 2, 1)
    ref = F.silu(F.group_norm(full, 32, gamma.cpu(), beta.cpu(), 1e-5)).permute(0, 2, 1)
    check_bf16(g_cols, ref)
    # the 256 x 320 tile (two staging passes per workgroup; chosen for long k loops on big grids): partials vs the column
    # sums of the tensor it stored, plus a ragged M tail
    for Bb, Hh in ((48, 32), (47, 32)):
        Mb = Bb * Hh * Hh
        xb = torch.randn(Bb, Hh, Hh, 256, device="cuda").to(BF)
        wb = (torch.randn(320, 9 * 256, device="cuda") * (9 * 256) ** -0.5).to(BF)
        cb = torch.zeros((Mb + 127) // 128, 320, 2, device="cuda")
        yb = ops.gemm(xb, wb, conv=dict(Hs=Hh, Ws=Hh, Cin=256, Ho=Hh, Wo=Hh), col_stats=cb)
        yf = yb.float()
        pad = (-Mb) % 128
        if pad:
            yf = torch.cat([yf, torch.zeros(pad, 320, device="cuda")])
        yf = yf.reshape(-1, 128, 320)
        torch.testing.assert_close(cb[..., 0], yf.sum(1), rtol=1e-5, atol=2e-3)
        torch.testing.assert_close(cb[..., 1], (yf * yf).sum(1), rtol=1e-5, atol=2e-3)


def test_conv_out_padded_n_to_f32_nchw():
    B, H, Cin, Cout = 2, 16, 320, 4
    x = rnd(B, Cin, H, H, seed=18)
    w = rnd(Cout, Cin, 3, 3, seed=19, scale=(9 * Cin) ** -0.5)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(20))
    ref = F.conv2d(x.float(), w.float(), bias, padding=1)
    wp = torch.zeros(32, 9 * Cin, dtype=BF); wp[:Cout] = w.permute(0, 2, 3, 1).reshape(Cout, -1)
    bp = torch.zeros(32); bp[:Cout] = bias
    out = ops.gemm(x.permute(0, 2, 3, 1).contiguous().cuda(), wp.cuda(), bias=bp.cuda(),
                   conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H), out_kind=2, n_valid=Cout)
    torch.testing.assert_close(out.reshape(B, Cout, H, H).cpu(), ref, rtol=2e-4, atol=2e-4)


def test_gemm_rejects_bad_shapes():
    a, w = rnd(64, 96).cuda(), rnd(32, 96).cuda()
    with pytest.raises(sda.SdnError):
        ops.gemm(a, w)                                                                   # K % 64 != 0
    a, w = rnd(64, 64).cuda(), rnd(48, 64).cuda()
    with pytest.raises(sda.SdnError):
        ops.gemm(a, w)                                                                   # N % 32 != 0


# ------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("B,hw,c1,c2,silu,eps", [(2, 4096, 320, 0, 1, 1e-5), (3, 256, 1280, 640, 1, 1e-5),
                                                 (1, 1024, 640, 320, 1, 1e-5), (2, 64, 1280, 1280, 1, 1e-5),
                                                 (2, 1024, 640, 0, 0, 1e-6), (1, 64, 64, 0, 1, 1e-5)])
def test_groupnorm_silu_concat(B, hw, c1, c2, silu, eps):
    x = rnd(B, hw, c1, seed=21) * 2 + 0.5
    x2 = (rnd(B, hw, c2, seed=22) * 0.5 - 1) if c2 else None
    C_ = c1 + c2
    g = 1 + 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(23))
    b = 0.1 * torch.randn(C_, generator=torch.Generator().manual_seed(24))
    full = torch.cat([x, x2], 2) if c2 else x
    ref = F.group_norm(full.float().permute(0, 2, 1), 32, g, b, eps=eps)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1)
    out = ops.groupnorm(x.cuda(), None if x2 is None else x2.cuda(), 32, eps, silu, g.cuda(), b.cuda())
    check_bf16(out, ref)


@pytest.mark.parametrize("rows,C", [(4096, 320), (1000, 640), (7, 1280), (64, 1536)])
def test_layernorm(rows, C):
    x = rnd(rows, C, seed=25) * 3 + 1
    g = 1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(26))
    b = 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(27))
    ref = F.layer_norm(x.float(), (C,), g, b, eps=1e-5)
    check_bf16(ops.layernorm(x.cuda(), g.cuda(), b.cuda()), ref)


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("B,H,Nq,Nk,d", [(2, 8, 256, 256, 40), (1, 8, 1024, 1024, 80), (2, 8, 64, 64, 160),
                                         (2, 8, 256, 77, 40), (1, 8, 64, 77, 160), (1, 4, 128, 200, 64),
                                         (1, 8, 4096, 4096, 40), (1, 8, 300, 300, 40),   # d = 40, long keys: two query sets per wave (+ ragged tails)
                                         (8, 8, 300, 77, 40), (16, 8, 128, 77, 80)])   # batch % 8 == 0, short keys: head-innermost block order
def test_attention_vs_sdpa(B, H, Nq, Nk, d):
    q, k, v = rnd(B, Nq, H * d, seed=28), rnd(B, Nk, H * d, seed=29), rnd(B, Nk, H * d, seed=30)
    sp = lambda t, n: t.float().reshape(B, n, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, Nq), sp(k, Nk), sp(v, Nk)).transpose(1, 2).reshape(B, Nq, H * d)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    # P is rounded to bf16 before the PV product (as every bf16 flash kernel does): 6e-3 relative L2
    assert rel_l2(out, ref) <= 6e-3, rel_l2(out, ref)


def test_attention_fused_qkv_views_and_sharp_softmax():
    """Strided q/k/v views of one [B,N,3C] projection; one key dominating (forces the running-max rescale path)."""
    B, H, N, d = 1, 8, 256, 40
    C_ = H * d
    qkv = rnd(B, N, 3 * C_, seed=31)
    qkv[0, 200, C_:2 * C_] = qkv[0, 5, :C_] * 6                      # key 200 matches query 5 strongly (late tile)
    q, k, v = qkv[..., :C_], qkv[..., C_:2 * C_], qkv[..., 2 * C_:]
    sp = lambda t: t.float().reshape(B, N, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, C_)
    g = qkv.cuda()
    out = ops.attention(g[..., :C_], g[..., C_:2 * C_], g[..., 2 * C_:], H)
    assert rel_l2(out, ref) <= 6e-3


@pytest.mark.parametrize("kind", ["spike_up_late", "all_far_down", "up_then_down"])
def test_attention_offset_recentering_branches(kind):
    """bf16 attention runs the softmax WITHOUT max subtraction while every tile maximum stays within 2^+-64 of the
    per-query offset; these inputs force the rare re-centring branch (guide rule 26: a data-dependent branch needs an
    input that takes it): logits far above the window in a late tile, far below it everywhere, and both in turn."""
    B, H, N, d = 1, 8, 512, 40
    C_ = H * d
    g = torch.Generator().manual_seed(35)
    q = torch.randn(B, N, C_, generator=g) * 0.5
    k = torch.randn(B, N, C_, generator=g) * 0.5
    v = torch.randn(B, N, C_, generator=g)
    if kind in ("spike_up_late", "up_then_down"):
        q[0, 7, :d] = 6.0; k[0, 450, :d] = 6.0                       # head 0, query 7 x key 450: logit = 36*40/sqrt(40) = 228 nats
    if kind in ("all_far_down", "up_then_down"):
        q[0, 9, d:2 * d] = 8.0; k[0, :, d:2 * d] = -3.0 + 0.05 * torch.randn(N, d, generator=g)   # head 1, query 9: all ~ -150 nats
    if kind == "up_then_down":
        k[0, 100, :d] = 4.0                                            # an earlier, smaller spike for query 7 (two re-centres)
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    sp = lambda t: t.float().reshape(B, N, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, C_)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) <= 8e-3, rel_l2(out, ref)
    # the forced rows themselves
    for row, hd in ((7, 0), (9, 1)):
        sl = slice(hd * d, (hd + 1) * d)
        assert rel_l2(out[0, row, sl], ref[0, row, sl]) <= 2e-2


@pytest.mark.parametrize("kind", ["plain", "spike_up_late", "all_far_down", "window_edge"])
@pytest.mark.parametrize("N,Nk,d", [(512, 512, 40), (256, 256, 80), (128, 128, 160), (256, 77, 40)])
def test_attention_optimistic_pass_gives_the_guarded_pass_bits(kind, N, Nk, d):
    """bf16 attention first runs WITHOUT the per-tile running maximum and checks the row sums afterwards (2^-64 <= sum <= 2^64);
    a workgroup with a query outside that window re-runs its block with the guarded loop.  Ordinary inputs: one pass, and
    the same bits as the guarded loop alone (sdn_debug_set_attn_nomax(0)); inputs that overflow / underflow 2^s: the re-run
    must give exactly what the guarded loop alone gives; logits at the edge of the window (2^+-40) stay on the fast pass and
    must still match the reference."""
    B, H = 1, 8
    C_ = H * d
    g = torch.Generator().manual_seed(41)
    q = torch.randn(B, N, C_, generator=g) * 0.5
    k = torch.randn(B, Nk, C_, generator=g) * 0.5
    v = torch.randn(B, Nk, C_, generator=g)
    s = d ** 0.5
    if kind == "spike_up_late":
        q[0, 7, :d] = 6.0 * (40 / d) ** 0.25; k[0, Nk - 30, :d] = 6.0 * (40 / d) ** 0.25      # ~228 nats: 2^s overflows
    if kind == "all_far_down":
        q[0, 9, d:2 * d] = 8.0 * (40 / d) ** 0.25
        k[0, :, d:2 * d] = (-3.0 + 0.05 * torch.randn(Nk, d, generator=g)) * (40 / d) ** 0.25     # all ~ -150 nats: every 2^s is 0
    if kind == "window_edge":
        q[0, 7, :d] = 2.0 * (40 / d) ** 0.25; k[0, Nk - 30, :d] = 2.1 * (40 / d) ** 0.25      # ~ +27 nats = 2^38
        q[0, 9, d:2 * d] = 2.0 * (40 / d) ** 0.25; k[0, :, d:2 * d] = -2.0 * (40 / d) ** 0.25  # all ~ -25 nats = 2^-36
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    sp = lambda t, n: t.float().reshape(B, n, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, N), sp(k, Nk), sp(v, Nk)).transpose(1, 2).reshape(B, N, C_)
    lib = sda.lib()
    try:
        lib.sdn_debug_set_attn_nomax(0)
        guarded = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
        torch.cuda.synchronize()
    finally:
        lib.sdn_debug_set_attn_nomax(1)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    assert torch.isfinite(out).all()
    assert torch.equal(out, guarded)
    assert rel_l2(out, ref) <= 8e-3, rel_l2(out, ref)


@pytest.mark.parametrize("kind", ["plain", "spike_up_late", "spike_up_moderate", "all_far_down", "ragged_cross"])
@pytest.mark.parametrize("d", [40, 64, 80, 160])
def test_attention_fp16_centred_pass_and_its_guarded_rerun(kind, d):
    """fp16 storage, d = 40 (round 5): the scale rides in Q, and each query's exponent is centred ONCE on its first tile's maximum m0
    -- column 40 of every K row reads 1.0, column 40 of the Q fragment is set to -m0 after tile 0, so the MFMA delivers s - m0 for
    the later tiles with no per-score VALU work and no running maximum; sum_k p <= 2^15 is checked afterwards and a workgroup
    with a query outside the window re-runs its block with the guarded loop.  Inputs: ordinary; a late key 2^20 above the first
    tile's maximum (p would overflow fp16: the re-run); one 2^9 above it (stays on the fast pass with p up to 512); every score far
    below zero (the offset does its job); 77 keys with a ragged tile.  Against float64 softmax attention on the fp16 values, and
    against the guarded loop alone (sdn_debug_set_attn_nomax(0)).  d = 64 / 80 / 160 have no free column: the same pass with the
    subtraction spelled out (one v_sub per score, still no running maximum)."""
    B, H, N = 1, 8, 512
    Nk = 77 if kind == "ragged_cross" else 512
    C_ = H * d
    g = torch.Generator().manual_seed(43)
    q = torch.randn(B, N, C_, generator=g) * 0.5
    k = torch.randn(B, Nk, C_, generator=g) * 0.5
    v = torch.randn(B, Nk, C_, generator=g)
    f = (40 / d) ** 0.25
    if kind == "spike_up_late":
        q[0, 7, :d] = 4.0 * f; k[0, Nk - 30, :d] = 4.0 * f                    # logit 16 * 40 / sqrt(40) = 101 nats = 2^146 above the rest
    if kind == "spike_up_moderate":
        q[0, 7, :d] = 1.0 * f; k[0, Nk - 30, :d] = 1.0 * f                    # 6.3 nats = 2^9: inside the window
    if kind == "all_far_down":
        q[0, 9, d:2 * d] = 4.0 * f; k[0, :, d:2 * d] = (-2.0 + 0.05 * torch.randn(Nk, d, generator=g)) * f     # all ~ -50 nats
    q, k, v = q.half(), k.half(), v.half()
    sp = lambda t, n: t.double().reshape(B, n, H, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q, N), sp(k, Nk), sp(v, Nk)).transpose(1, 2).reshape(B, N, C_)
    lib = sda.lib()
    try:
        lib.sdn_debug_set_attn_nomax(0)
        guarded = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
        torch.cuda.synchronize()
    finally:
        lib.sdn_debug_set_attn_nomax(1)
    out = ops.attention(q.cuda(), k.cuda(), v.cuda(), H)
    assert torch.isfinite(out).all()
    r_out, r_g, r_between = rel_l2(out, ref), rel_l2(guarded, ref), rel_l2(out, guarded)
    print(f"fp16 attention d={d} {kind}: vs float64 {r_out:.2e} (guarded loop alone {r_g:.2e}); between the two {r_between:.2e}")
    assert r_out <= 1.2e-3 and r_g <= 1.2e-3                                  # P and the output are rounded to fp16: 2^-11 per element
    for row, hd in ((7, 0), (9, 1)):                                          # the forced rows themselves
        sl = slice(hd * d, (hd + 1) * d)
        assert rel_l2(out[0, row, sl], ref[0, row, sl]) <= 3e-3
    if kind == "spike_up_late":
        # the overflowing block: its re-run IS the guarded loop -> the same bits for that workgroup's rows (256 queries at d = 40:
        # two query sets per wave; 128 otherwise)
        blk = slice(0, 256 if d == 40 else 128)
        assert torch.equal(out[0, blk, :d], guarded[0, blk, :d])
    else:
        assert r_between <= 1.2e-3 and (kind == "ragged_cross" or not torch.equal(out, guarded))       # two different (valid) offsets


# ------------------------------------------------------------------------------------------ conv_in / temb
def test_conv_in_and_timestep_embedding():
    B, H = 2, 64
    lat = torch.randn(B, 4, H, H, generator=torch.Generator().manual_seed(32))
    w = rnd(320, 4, 3, 3, seed=33, scale=1 / 6)
    bias = torch.randn(320, generator=torch.Generator().manual_seed(34))
    ref = F.conv2d(lat, w.float(), bias, padding=1).permute(0, 2, 3, 1)
    out = torch.empty(B, H, H, 320, dtype=BF, device="cuda")
    wn, lg, bg = w.permute(0, 2, 3, 1).contiguous().cuda(), lat.cuda(), bias.cuda()     # keep device tensors alive
    _lib.check(sda.lib().sdn_conv_in_bf16(lg.data_ptr(), wn.data_ptr(), bg.data_ptr(), B, 4, H, H, 320,
                                          out.data_ptr(), _lib.stream_ptr()), "conv_in")
    torch.cuda.synchronize()
    check_bf16(out, ref)
    # odd sizes: width not a multiple of the 4-pixel thread tile, 3 input channels, few output channels
    for (B2, cin, H2, W2, cout) in ((1, 3, 5, 7, 16), (2, 16, 9, 13, 40)):
        lat2 = torch.randn(B2, cin, H2, W2, generator=torch.Generator().manual_seed(35))
        w2 = rnd(cout, cin, 3, 3, seed=36, scale=(9 * cin) ** -0.5)
        b2 = torch.randn(cout, generator=torch.Generator().manual_seed(37))
        ref2 = F.conv2d(lat2, w2.float(), b2, padding=1).permute(0, 2, 3, 1)
        o2 = torch.empty(B2, H2, W2, cout, dtype=BF, device="cuda")
        wn2, lg2, bg2 = w2.permute(0, 2, 3, 1).contiguous().cuda(), lat2.cuda(), b2.cuda()
        _lib.check(sda.lib().sdn_conv_in_bf16(lg2.data_ptr(), wn2.data_ptr(), bg2.data_ptr(), B2, cin, H2, W2, cout,
                                              o2.data_ptr(), _lib.stream_ptr()), "conv_in")
        check_bf16(o2, ref2)
    from oracle.unet import OracleUNet
    for t in (981.0, 1.0, 500.0):
        te = torch.empty(3, 320, dtype=BF, device="cuda")
        _lib.check(sda.lib().sdn_timestep_embed_bf16(t, 3, 320, te.data_ptr(), _lib.stream_ptr()), "temb")
        exp = OracleUNet({}, None).timestep_features(t, 3, 320)
        assert float((te.float().cpu() - exp).abs().max()) <= 2 ** -8 + 2e-4     # bf16 rounding + fp32 sin/cos argument


# ------------------------------------------------------------------------------------------ fp16 storage twins
def test_f16_twins_are_8x_tighter():
    """The _f16 entry points (IEEE half storage, same kernels through the dtype traits): one output rounding is 2^-12
    relative -> rel L2 <= 5e-4 (attention 1.5e-3: P is rounded to fp16 before the PV product)."""
    H16 = torch.float16
    g = torch.Generator().manual_seed(40)
    a = torch.randn(512, 640, generator=g).to(H16); w = (torch.randn(320, 640, generator=g) * 640 ** -0.5).to(H16)
    bias = torch.randn(320, generator=g)
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda()), a.float() @ w.float().T + bias) <= 5e-4
    x = torch.randn(2, 16, 16, 320, generator=g).to(H16); cw = (torch.randn(320, 320, 3, 3, generator=g) / 54).to(H16)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), cw.float(), None, padding=1)
    out = ops.gemm(x.cuda(), cw.permute(0, 2, 3, 1).reshape(320, -1).contiguous().cuda(),
                   conv=dict(Hs=16, Ws=16, Cin=320, Ho=16, Wo=16))
    assert rel_l2(out.reshape(2, 16, 16, 320).permute(0, 3, 1, 2), ref) <= 5e-4
    xx = (torch.randn(2, 256, 640, generator=g) * 2 + 0.5).to(H16)
    gm = 1 + 0.1 * torch.randn(640, generator=g); bt = 0.1 * torch.randn(640, generator=g)
    ref = F.silu(F.group_norm(xx.float().permute(0, 2, 1), 32, gm, bt, eps=1e-5)).permute(0, 2, 1)
    assert rel_l2(ops.groupnorm(xx.cuda(), None, 32, 1e-5, 1, gm.cuda(), bt.cuda()), ref) <= 5e-4
    ref = F.layer_norm(xx[0].float(), (640,), gm, bt, eps=1e-5)
    assert rel_l2(ops.layernorm(xx[0].contiguous().cuda(), gm.cuda(), bt.cuda()), ref) <= 5e-4
    q, k, v = (torch.randn(1, 256, 320, generator=g).to(H16) for _ in range(3))
    sp = lambda t: t.float().reshape(1, 256, 8, 40).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(1, 256, 320)
    assert rel_l2(ops.attention(q.cuda(), k.cuda(), v.cuda(), 8), ref) <= 1.5e-3


def test_groupnorm_from_column_sums_with_a_large_mean():
    """E[x^2] - mean^2 from fp32 column sums of 16-bit values (sdn_groupnorm_cols_*) when |mean| >> std: the variance then
    loses log2((mean/std)^2) bits to cancellation.  mean/std = 100 (far beyond what the UNet's activations show: their
    |mean|/std stays below ~3) must still be within the 16-bit output rounding of torch's GroupNorm; the bias term comes
    from a GEMM bias so that the producing kernel really is the GEMM."""
    B, hw, K, N = 2, 256, 64, 320
    M = B * hw
    a = rnd(M, K, seed=91); w = rnd(N, K, seed=92, scale=K ** -0.5)
    bias = torch.full((N,), 100.0)                                       # y = N(0, 1) + 100
    cols = torch.zeros(M // 128, N, 2, device="cuda")
    y = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), col_stats=cols)
    gamma = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(93))).cuda()
    beta = (0.1 * torch.randn(N, generator=torch.Generator().manual_seed(94))).cuda()
    g_cols = ops.groupnorm(y.reshape(B, hw, N), None, 32, 1e-5, 0, gamma, beta, cols1=cols)
    g_plain = ops.groupnorm(y.reshape(B, hw, N), None, 32, 1e-5, 0, gamma, beta)
    ref = F.group_norm(y.float().cpu().reshape(B, hw, N).permute(0, 2, 1).double(), 32, gamma.cpu().double(), beta.cpu().double(), 1e-5)
    ref = ref.permute(0, 2, 1)
    r_cols, r_plain = rel_l2(g_cols, ref), rel_l2(g_plain, ref)
    print(f"GroupNorm at mean/std = 100: rel L2 vs float64, from column sums {r_cols:.2e}, two-pass kernel {r_plain:.2e}")
    assert r_cols <= 6e-3 and r_plain <= 6e-3


def test_geglu_epilogue_gelu_is_within_the_16_bit_output_rounding():
    """The GEGLU epilogue evaluates GELU with a quintic-argument logistic (<= 2.6e-5 absolute from the erf form).  With fp16
    storage -- the tighter of the two 16-bit modes -- the result must sit at one output rounding from torch's exact-erf GEGLU,
    and gates spanning the whole useful range (|g| up to 12) must not show the approximation."""
    from safe_denoiser_amd.unet import _interleave16
    M, C = 512, 320
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, C, generator=g) * 1.5).half()
    w = (torch.randn(8 * C, C, generator=g) * C ** -0.5 * 2.0).half()           # gates with std ~3: tails well exercised
    b = torch.randn(8 * C, generator=g)
    proj = x.float() @ w.float().T + b
    val, gate = proj.chunk(2, -1)
    ref = val * F.gelu(gate)
    out = ops.gemm(x.cuda(), _interleave16(w).contiguous().cuda(), bias=_interleave16(b).contiguous().cuda(), act=2)
    err = (out.float().cpu() - ref)
    r = float(err.norm() / ref.norm())
    print(f"GEGLU f16: rel L2 {r:.2e}, gate range [{float(gate.min()):.1f}, {float(gate.max()):.1f}]")
    assert r <= 4e-4                                                            # fp16 rounding alone: ~2.9e-4
    assert float(gate.abs().max()) > 10.0


# ------------------------------------------------------------------------------------------ fused GEGLU feed-forward
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [128 * 5 + 37, 4096 * 6])
def test_fused_geglu_feed_forward_is_bit_identical_to_the_two_launches(dt, M):
    """sdn_ffn_geglu_fused (csrc/sdn_ffn.hip): LayerNorm-folded GEGLU projection + the [ff | h3] . [Wpo W2 | Wpo]^T contraction
    + residual + GroupNorm column sums with the hidden activation kept in LDS.  Same k order, same epilogue expressions ->
    the SAME BITS as sdn_gemm_ln_* (GEGLU, pre-pass statistics) followed by the two-source sdn_gemm_stats_*, including a
    ragged last row block; and within 16-bit rounding of plain fp32 arithmetic."""
    from safe_denoiser_amd.unet import _interleave16
    Cc = 320
    g = torch.Generator().manual_seed(11)
    x = ((torch.randn(M, Cc, generator=g) * 1.5 + torch.randn(M, 1, generator=g) * 2.0)).to(dt).cuda()
    w1 = (torch.randn(8 * Cc, Cc, generator=g) * Cc ** -0.5).to(dt)
    b1 = torch.randn(8 * Cc, generator=g)
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g); beta = 0.3 * torch.randn(Cc, generator=g)
    wcat = (torch.randn(Cc, 5 * Cc, generator=g) * (5 * Cc) ** -0.5).to(dt).cuda()
    bcat = torch.randn(Cc, generator=g).cuda()
    res = torch.randn(M, Cc, generator=g).to(dt).cuda()
    w1i, b1i = _interleave16(w1).contiguous().cuda(), _interleave16(b1).contiguous().cuda()
    nblk = (M + 127) // 128
    cs_a = torch.zeros(nblk, Cc, 2, device="cuda"); cs_b = torch.zeros_like(cs_a)
    ff = ops.gemm_ln(x, w1i, gamma.cuda(), beta.cuda(), b1i, act=2, prepass=True)
    want = ops.gemm(ff, wcat, a2=x, bias=bcat, residual=res, col_stats=cs_a)
    got = ops.ffn_fused(x, w1i, gamma.cuda(), beta.cuda(), b1i, wcat, bcat, res, col_stats=cs_b)
    torch.cuda.synchronize()
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), float((got.float() - want.float()).abs().max())
    assert torch.equal(cs_a, cs_b)
    # fp32 arithmetic on the same 16-bit operands (hidden activation NOT rounded): bounded by the 16-bit roundings
    y = F.linear(F.layer_norm(x.float().cpu(), (Cc,), gamma, beta, 1e-5), w1.float(), b1)
    hid = y[:, :4 * Cc] * F.gelu(y[:, 4 * Cc:])
    ref = res.float().cpu() + torch.cat([hid, x.float().cpu()], 1) @ wcat.float().cpu().T + bcat.cpu()
    assert rel_l2(got, ref) <= (6e-3 if dt == torch.bfloat16 else 8e-4)
    # row_stats = NULL: norm3's statistics from the kernel's own operand fragments = the bits of sdn_gemm_ln_* WITHOUT a pre-pass
    # (fragment statistics, same dot-product order), followed by the same two-source GEMM
    ff1 = ops.gemm_ln(x, w1i, gamma.cuda(), beta.cuda(), b1i, act=2, prepass=False)
    want1 = ops.gemm(ff1, wcat, a2=x, bias=bcat, residual=res)
    got1 = ops.ffn_fused(x, w1i, gamma.cuda(), beta.cuda(), b1i, wcat, bcat, res, own_stats=True)
    assert torch.equal(got1.view(torch.int16), want1.view(torch.int16)), float((got1.float() - want1.float()).abs().max())
    assert rel_l2(got1, ref) <= (6e-3 if dt == torch.bfloat16 else 8e-4)
    # no column sums requested; other widths are refused (the caller keeps the two-launch form)
    assert torch.equal(ops.ffn_fused(x, w1i, gamma.cuda(), beta.cuda(), b1i, wcat, bcat, res).view(torch.int16), want.view(torch.int16))
    rc = sda.lib().sdn_ffn_geglu_fused(0, 128, 640, x.data_ptr(), cs_a.data_ptr(), w1i.data_ptr(), b1i.data_ptr(), b1i.data_ptr(),
                                       wcat.data_ptr(), bcat.data_ptr(), res.data_ptr(), got.data_ptr(), None, None)
    assert rc == sda.SDN_E_INVALID if hasattr(sda, "SDN_E_INVALID") else rc != 0


# ------------------------------------------------------------------------------------------ tile choice for short k loops
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_big_tile_on_short_k_projections_gives_the_small_tile_bits(dt):
    """Round 2 moved the short-K projections whose epilogue only computes and stores (qkv, proj_in, GEGLU, LayerNorm-folded
    forms) onto the 256 x 320 tile (sdn_gemm_pick_tile; variant 9 = the round-1 rule).  The tile must not change a single bit
    (same k order per output element), whatever the epilogue: plain, bias, LayerNorm fold with fragment or pre-pass statistics,
    GEGLU; a residual keeps the small tile either way.  Also checked against fp32 arithmetic."""
    from safe_denoiser_amd.unet import _interleave16
    g = torch.Generator().manual_seed(21)
    t = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dt).cuda()
    M = 256 * 70 + 19                                                # 71 row tiles of 256, ragged tail
    a3, a6 = t(M, 320), t(M, 640)
    gam3 = (1 + 0.2 * torch.randn(320, generator=g)).cuda(); bet3 = (0.3 * torch.randn(320, generator=g)).cuda()
    gam6 = (1 + 0.2 * torch.randn(640, generator=g)).cuda(); bet6 = (0.3 * torch.randn(640, generator=g)).cuda()
    w960, b960 = t(960, 320, scale=320 ** -0.5), torch.randn(960, generator=g).cuda()
    w1920, b1920 = t(1920, 640, scale=640 ** -0.5), torch.randn(1920, generator=g).cuda()
    wg = _interleave16(t(2560, 320, scale=320 ** -0.5)).contiguous(); bg = _interleave16(torch.randn(2560, generator=g).cuda()).contiguous()
    wg6 = _interleave16(t(5120, 640, scale=640 ** -0.5)).contiguous(); bg6 = _interleave16(torch.randn(5120, generator=g).cuda()).contiguous()
    res = t(M, 960)
    cases = [("qkv 320 plain", lambda: ops.gemm(a3, w960, bias=b960)),
             ("qkv 320 LN fold, fragment statistics", lambda: ops.gemm_ln(a3, w960, gam3, bet3, b960, prepass=False)),
             ("qkv 640 LN fold, pre-pass statistics", lambda: ops.gemm_ln(a6, w1920, gam6, bet6, b1920, prepass=True)),
             ("GEGLU 320 LN fold", lambda: ops.gemm_ln(a3, wg, gam3, bet3, bg, act=2, prepass=True)),
             ("GEGLU 640", lambda: ops.gemm(a6, wg6, bias=bg6, act=2)),
             ("qkv 320 + residual (small tile either way)", lambda: ops.gemm(a3, w960, bias=b960, residual=res))]
    try:
        for name, fn in cases:
            _variant(9); want = fn(); torch.cuda.synchronize()
            _variant(0); got = fn(); torch.cuda.synchronize()
            assert torch.isfinite(got.float()).all(), name
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (name, float((got.float() - want.float()).abs().max()))
    finally:
        _variant(0)
    ref = (a3.float() @ w960.float().T + b960).cpu()
    assert rel_l2(ops.gemm(a3, w960, bias=b960), ref) <= (4e-3 if dt == torch.bfloat16 else 6e-4)
    y = F.linear(F.layer_norm(a6.float().cpu(), (640,), gam6.cpu(), bet6.cpu(), 1e-5), w1920.float().cpu(), b1920.cpu())
    assert rel_l2(ops.gemm_ln(a6, w1920, gam6, bet6, b1920, prepass=True), y) <= (6e-3 if dt == torch.bfloat16 else 8e-4)


def _variant(v):
    sda.lib().sdn_debug_set_gemm_variant(int(v))


# ------------------------------------------------------------------------------------------ slab-ring 3x3 convolution
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,Cin,Cout", [(13, 64, 320, 320), (12, 64, 640, 640), (26, 32, 640, 640), (48, 32, 320, 320),
                                          (52, 16, 1280, 1280)])
def test_slab_ring_convolution_gives_the_implicit_gemm_bits(dt, B, H, Cin, Cout):
    """csrc/sdn_conv.hip: stride-1 3x3 convs on 64 / 32 / 16-wide maps read their A fragments from an LDS slab ring (the input
    window DMA'd once per 64-channel chunk) instead of re-fetching a shifted A tile per tap.  Same k order, same MFMAs, same
    epilogue -> the SAME BITS as the implicit-GEMM kernel (variant 13 switches the slab form off): plain, + per-sample row bias
    (time embedding), + residual, + GroupNorm column sums; image borders (zero padding on all four sides), several samples per
    launch, 5 ... 20 channel chunks through the ring.  And against torch's fp32 conv2d."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, H, H, Cin, generator=g).to(dt).cuda()
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) * (9 * Cin) ** -0.5).to(dt).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    rowbias = torch.randn(B, Cout, generator=g).cuda()
    res = torch.randn(B * H * H, Cout, generator=g).to(dt).cuda()
    conv = dict(Hs=H, Ws=H, Cin=Cin, Ho=H, Wo=H)
    w2 = w.reshape(Cout, 9 * Cin)
    nblk = (B * H * H + 127) // 128
    cs_a = torch.zeros(nblk, Cout, 2, device="cuda"); cs_b = torch.zeros_like(cs_a)
    cases = [("plain", lambda cs: ops.gemm(x, w2, bias=bias, conv=conv)),
             ("row bias + residual", lambda cs: ops.gemm(x, w2, bias=bias, rowbias=rowbias, residual=res, conv=conv)),
             ("residual + column sums", lambda cs: ops.gemm(x, w2, bias=bias, residual=res, conv=conv, col_stats=cs))]
    import ctypes
    counts = (ctypes.c_longlong * 2)()

    def launched(reset=True):
        sda.lib().sdn_debug_gemm_launch_counts(counts, 1 if reset else 0)
        return counts[0], counts[1]
    try:
        for name, fn in cases:
            launched()
            _variant(13); want = fn(cs_a); torch.cuda.synchronize()
            assert launched() == (0, 1), (name, "variant 13 must take the implicit-GEMM kernel")
            _variant(0); got = fn(cs_b); torch.cuda.synchronize()
            assert launched() == (1, 0), (name, "the default build must take the slab-ring kernel for this shape")
            assert torch.isfinite(got.float()).all(), name
            assert torch.equal(got.view(torch.int16), want.view(torch.int16)), (name, float((got.float() - want.float()).abs().max()))
        assert torch.equal(cs_a, cs_b)
    finally:
        _variant(0)
    ref = F.conv2d(x[:2].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    out = ops.gemm(x, w2, bias=bias, conv=conv)[:2 * H * H]
    assert rel_l2(out, ref.cpu()) <= (4e-3 if dt == torch.bfloat16 else 6e-4)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_residual_into_the_accumulators_matches_the_epilogue_residual_and_is_tile_invariant(dt):
    """sdn_gemm_desc.res_pre (round 4): the 16-bit residual of an attention output projection / conv2 / FeedForward-output GEMM is
    added into the ACCUMULATORS before the k loop (its loads ride under the first k-tile's DMA; the epilogue is the lean one) instead
    of through the epilogue's staging slab.  The sum is formed in another order -- equal to the epilogue form to one output rounding,
    and to the fp32 reference within the 16-bit bound -- and the SAME BITS come out of every tile family / the slab-ring convolution
    (variant 3 = 128-row tile, 13 = slab off)."""
    g = torch.Generator().manual_seed(41)
    tol = 4e-3 if dt == torch.bfloat16 else 6e-4
    # plain GEMM (attention to_out at C = 320 / 640): big tile vs 128-row tile
    for M, N, K in ((4096 * 3, 320, 320), (3000, 640, 640)):
        a = torch.randn(M, K, generator=g).to(dt).cuda(); w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).cuda()
        bias = torch.randn(N, generator=g).cuda(); res = torch.randn(M, N, generator=g).to(dt).cuda()
        ref = a.float() @ w.float().T + bias + res.float()
        try:
            _variant(0); pre = ops.gemm(a, w, bias=bias, residual=res, res_pre=1)
            _variant(3); pre_small = ops.gemm(a, w, bias=bias, residual=res, res_pre=1)
        finally:
            _variant(0)
        epi = ops.gemm(a, w, bias=bias, residual=res)
        assert torch.equal(pre.view(torch.int16), pre_small.view(torch.int16))
        assert rel_l2(pre, ref.cpu()) <= tol and rel_l2(epi, ref.cpu()) <= tol
        # at most one output rounding apart at the magnitude of the summands (~1): where an output is a near-cancellation the two
        # fp32 sums differ by the accumulation noise of the larger terms, a few of the small result's own ulps
        ulp = (pre.float() - epi.float()).abs() / epi.float().abs().clamp_min(1.0)
        assert float(ulp.max()) <= (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * 1.01
    # conv2 of a resnet (+ GroupNorm column sums): slab-ring kernel vs implicit GEMM
    B, H, C_ = 48, 32, 320
    x = torch.randn(B, H, H, C_, generator=g).to(dt).cuda()
    w2 = (torch.randn(C_, 9 * C_, generator=g) * (9 * C_) ** -0.5).to(dt).cuda()
    bias = torch.randn(C_, generator=g).cuda(); res = torch.randn(B * H * H, C_, generator=g).to(dt).cuda()
    conv = dict(Hs=H, Ws=H, Cin=C_, Ho=H, Wo=H)
    nblk = (B * H * H + 127) // 128
    cs_a = torch.zeros(nblk, C_, 2, device="cuda"); cs_b = torch.zeros_like(cs_a)
    try:
        _variant(13); want = ops.gemm(x, w2, bias=bias, residual=res, conv=conv, col_stats=cs_a, res_pre=1)
        _variant(0); got = ops.gemm(x, w2, bias=bias, residual=res, conv=conv, col_stats=cs_b, res_pre=1)
    finally:
        _variant(0)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)) and torch.equal(cs_a, cs_b)
    epi = ops.gemm(x, w2, bias=bias, residual=res, conv=conv)
    assert float((got.float() - epi.float()).abs().max()) <= float(epi.float().abs().max()) * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
