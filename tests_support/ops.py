"""Thin ctypes callers of libsdn's operator-level entry points, shared by the GPU tests and the benchmark."""
import ctypes as C

import torch

import safe_denoiser_amd as sda
from safe_denoiser_amd import _lib

BF = torch.bfloat16


X3 = False      # tests of the bf16x3 contraction mode set this: f32 tensors then go to sdn_gemm_x3 / sdn_attention_x3


def _fn(base: str, t: torch.Tensor):
    """sdn_<base>_bf16, sdn_<base>_f16 or sdn_<base>_f32 (sdn_<base>_x3 under X3) by the tensor's storage dtype."""
    suffix = {torch.float16: "f16", torch.float32: "f32"}.get(t.dtype, "bf16")
    if X3 and suffix == "f32" and base in ("gemm", "attention"):
        suffix = "x3"
    return getattr(sda.lib(), f"sdn_{base}_{suffix}")


def gemm(a, w, *, bias=None, rowbias=None, residual=None, a2=None, conv=None, act=0, out_kind=0, n_valid=0,
         rows_per_batch=0, rowgate=None, residual_bcast=0, split_k=0, col_stats=None, res_pre=0):
    """a [M,K] bf16 (or NHWC map for conv=dict(Hs,Ws,Cin,Ho,Wo,stride,upsample)), w [N,K] bf16."""
    N, K = w.shape
    d = _lib.GemmDesc()
    if conv:
        B = a.shape[0]
        d.a_mode = 1
        d.Hs, d.Ws, d.Cin, d.Ho, d.Wo = conv["Hs"], conv["Ws"], conv["Cin"], conv["Ho"], conv["Wo"]
        d.stride, d.upsample = conv.get("stride", 1), conv.get("upsample", 0)
        d.asym_pad = conv.get("asym_pad", 0)
        M = B * d.Ho * d.Wo
        rows_per_batch = d.Ho * d.Wo
    else:
        M = a.shape[0]
        d.K1 = a.shape[1] if a2 is not None else 0
    d.M, d.N, d.K, d.act, d.out_kind, d.n_valid = M, N, K, act, out_kind, n_valid
    d.rows_per_batch = rows_per_batch
    d.res_pre = res_pre
    if rowbias is not None:
        d.ld_rowbias = rowbias.stride(0)
    if rowgate is not None:
        d.ld_rowgate = rowgate.stride(0)
    d.residual_bcast = residual_bcast
    nv = n_valid or N
    if out_kind == 0:
        out = torch.empty((M, N // 2 if act == 2 else nv), dtype=a.dtype, device=a.device)
    elif out_kind == 1:
        out = torch.empty((M, nv), dtype=torch.float32, device=a.device)
    else:
        out = torch.empty((M // rows_per_batch, nv, rows_per_batch), dtype=torch.float32, device=a.device)
    p = lambda t: None if t is None else t.data_ptr()
    if col_stats is not None:
        _lib.check(_fn("gemm_stats", a)(C.byref(d), p(a), p(a2), p(w), p(bias), p(rowbias), p(residual), p(out), p(col_stats),
                                                 _lib.stream_ptr()), "sdn_gemm_stats")
        return out
    if split_k > 1:
        d.split_k = split_k
        part = torch.empty(split_k * M * N, dtype=torch.float32, device=a.device)
        _lib.check(_fn("gemm_splitk", a)(C.byref(d), p(a), p(a2), p(w), p(bias), p(rowbias), p(rowgate), p(residual), p(out),
                                                  part.data_ptr(), part.numel() * 4, _lib.stream_ptr()), "sdn_gemm_splitk")
        return out
    _lib.check(_fn("gemm", a)(C.byref(d), p(a), p(a2), p(w), p(bias), p(rowbias), p(rowgate), p(residual), p(out),
                                       _lib.stream_ptr()), "sdn_gemm_bf16")
    return out


def groupnorm(x, x2, groups, eps, silu, gamma, beta, cols1=None, cols2=None):
    B, hw, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[2]
    out = torch.empty((B, hw, c1 + c2), dtype=x.dtype, device=x.device)
    ws = torch.empty(B * 129 * groups * 2, dtype=torch.float32, device=x.device)
    if cols1 is not None:
        _lib.check(_fn("groupnorm_cols", x)(x.data_ptr(), None if x2 is None else x2.data_ptr(), B, hw, c1, c2, groups, eps, silu,
                                                     gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), ws.data_ptr(), cols1.data_ptr(),
                                                     None if cols2 is None else cols2.data_ptr(), _lib.stream_ptr()), "sdn_groupnorm_cols")
        return out
    _lib.check(_fn("groupnorm", x)(x.data_ptr(), None if x2 is None else x2.data_ptr(), B, hw, c1, c2, groups,
                                            eps, silu, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                            ws.data_ptr(), _lib.stream_ptr()), "sdn_groupnorm_bf16")
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    rows, c = x.shape
    out = torch.empty_like(x)
    _lib.check(_fn("layernorm", x)(x.data_ptr(), rows, c, eps, gamma.data_ptr(), beta.data_ptr(),
                                            out.data_ptr(), _lib.stream_ptr()), "sdn_layernorm_bf16")
    return out


def attention(q, k, v, heads, scale=None):
    """q [B,Nq,H*d], k/v [B,Nk,H*d] bf16 (may be strided views of a fused projection)."""
    B, nq, c = q.shape
    nk = k.shape[1]
    d = c // heads
    out = torch.empty((B, nq, c), dtype=q.dtype, device=q.device)
    scale = scale if scale is not None else d ** -0.5
    _lib.check(_fn("attention", q)(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, heads, nq, nk,
                                            d, q.stride(1), k.stride(1), v.stride(1), c, scale, _lib.stream_ptr()),
               "sdn_attention_bf16")
    return out


def gemm_ln(x, w, gamma, beta, bias=None, act=0, eps=1e-5, prepass=False):
    """LayerNorm(x; gamma, beta) . w^T + bias [GEGLU] through sdn_ln_fold + sdn_gemm_ln_* (w [N, K] 16-bit, already in the
    layout the kernel expects -- interleaved for GEGLU)."""
    N, K = w.shape
    M = x.shape[0]
    code = 1 if x.dtype == torch.float16 else 0
    wf = torch.empty_like(w)
    c = torch.empty(N, dtype=torch.float32, device=x.device); dv = torch.empty_like(c)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(sda.lib().sdn_ln_fold(code, p(w), p(gamma), p(beta), p(bias), N, K, p(wf), p(c), p(dv), _lib.stream_ptr()), "sdn_ln_fold")
    d = _lib.GemmDesc()
    d.M, d.N, d.K, d.act = M, N, K, act
    out = torch.empty((M, N // 2 if act == 2 else N), dtype=x.dtype, device=x.device)
    stats = None
    if prepass:
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        _lib.check(_fn("row_stats", x)(p(x), M, K, eps, p(stats), _lib.stream_ptr()), "sdn_row_stats")
    _lib.check(_fn("gemm_ln", x)(C.byref(d), p(x), p(wf), p(c), p(dv), eps, p(stats), p(out), _lib.stream_ptr()), "sdn_gemm_ln")
    return out


def ffn_fused(x, w1, gamma, beta, bias1, w_cat, b_cat, residual, col_stats=None, eps=1e-5, own_stats=False):
    """sdn_ln_fold + sdn_row_stats_* + sdn_ffn_geglu_fused (w1 [8C, C] value / gate-interleaved, w_cat [C, 5C]).
    own_stats: row_stats = NULL, the kernel takes norm3's statistics from its own operand fragments."""
    N, K = w1.shape
    M = x.shape[0]
    code = 1 if x.dtype == torch.float16 else 0
    wf = torch.empty_like(w1)
    c = torch.empty(N, dtype=torch.float32, device=x.device); dv = torch.empty_like(c)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(sda.lib().sdn_ln_fold(code, p(w1), p(gamma), p(beta), p(bias1), N, K, p(wf), p(c), p(dv), _lib.stream_ptr()), "sdn_ln_fold")
    stats = None
    if not own_stats:
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        _lib.check(_fn("row_stats", x)(p(x), M, K, eps, p(stats), _lib.stream_ptr()), "sdn_row_stats")
    out = torch.empty((M, K), dtype=x.dtype, device=x.device)
    _lib.check(sda.lib().sdn_ffn_geglu_fused(code, M, K, p(x), p(stats), p(wf), p(c), p(dv), p(w_cat), p(b_cat), p(residual), p(out),
                                             p(col_stats), _lib.stream_ptr()), "sdn_ffn_geglu_fused")
    return out


# ---- bf16x3 by operand expansion (include/sdn.h): triples, expanded weights, sdn_gemm_bf16 with x3_out ------------------------
def split3(x, x2=None):
    """f32 [rows, c1] (++ [rows, c2]) -> bf16 triple [rows, 3 (c1 + c2)] = [hi | lo | hi]."""
    rows, c1 = x.shape
    c2 = 0 if x2 is None else x2.shape[1]
    out = torch.empty((rows, 3 * (c1 + c2)), dtype=BF, device=x.device)
    _lib.check(sda.lib().sdn_split3(x.data_ptr(), None if x2 is None else x2.data_ptr(), rows, c1, c2, out.data_ptr(),
                                    _lib.stream_ptr()), "sdn_split3")
    return out


def triple_value(t):
    """What a triple stands for: hi + lo, as float64 [rows, C] (and the check that its two hi planes agree)."""
    C3 = t.shape[-1]
    C_ = C3 // 3
    hi, lo, hi2 = t[..., :C_], t[..., C_:2 * C_], t[..., 2 * C_:]
    assert torch.equal(hi, hi2)
    return hi.double() + lo.double()


def expand3(w, group=None):
    """f32 W [N, K] -> bf16 [N, 3K], every `group` columns expanded to [hi | hi | lo]."""
    N, K = w.shape
    out = torch.empty((N, 3 * K), dtype=BF, device=w.device)
    _lib.check(sda.lib().sdn_expand3_weights(w.data_ptr(), N, K, group or K, out.data_ptr(), _lib.stream_ptr()), "sdn_expand3_weights")
    return out


def gemm_x3t(a3, w3, N, K, *, bias=None, rowbias=None, residual=None, conv=None, act=0, x3_out=1, rows_per_batch=0, out_kind=1,
             n_valid=0):
    """sdn_gemm_bf16 over a triple A operand [M, 3K] (or an NHWC triple map for conv) and an expanded weight [N, 3K]:
    K / Cin below are the LOGICAL sizes.  x3_out 1 -> f32 [M, N]; 2 (GEGLU) / 3 -> triple [M, 3 N'] ; 0 -> out_kind as given."""
    d = _lib.GemmDesc()
    if conv:
        B = a3.shape[0]
        d.a_mode = 1
        d.Hs, d.Ws, d.Cin, d.Ho, d.Wo = conv["Hs"], conv["Ws"], 3 * conv["Cin"], conv["Ho"], conv["Wo"]
        d.stride, d.upsample, d.asym_pad = conv.get("stride", 1), conv.get("upsample", 0), conv.get("asym_pad", 0)
        M = B * d.Ho * d.Wo
        rows_per_batch = d.Ho * d.Wo
    else:
        M = a3.shape[0]
    d.M, d.N, d.K, d.act, d.out_kind, d.x3_out, d.n_valid = M, N, 3 * K, act, out_kind, x3_out, n_valid
    d.rows_per_batch = rows_per_batch
    if rowbias is not None:
        d.ld_rowbias = rowbias.stride(0)
    width = N // 2 if act == 2 else N
    if x3_out == 1:
        out = torch.empty((M, width), dtype=torch.float32, device=a3.device)
    elif x3_out in (2, 3):
        out = torch.empty((M, 3 * width), dtype=BF, device=a3.device)
    elif x3_out == 4:
        out = torch.empty((M, 2 * width), dtype=BF, device=a3.device)
    else:
        out = torch.empty((M // rows_per_batch, n_valid or N, rows_per_batch), dtype=torch.float32, device=a3.device)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(sda.lib().sdn_gemm_bf16(C.byref(d), p(a3), None, p(w3), p(bias), p(rowbias), None, p(residual), p(out),
                                       _lib.stream_ptr()), "sdn_gemm_bf16 (x3_out)")
    return out


def attention_x3_pairs(qkv_pairs, heads, triple=False, scale=None):
    """qkv_pairs [B, N, 2 * 3C] bf16: rows [hi(q | k | v) | lo(q | k | v)] (sdn_gemm_bf16 with x3_out = 4) -> f32 [B, N, C]
    or the triple [B, N, 3C]."""
    B, n, w = qkv_pairs.shape
    c = w // 6
    d = c // heads
    out = torch.empty((B, n, 3 * c), dtype=BF, device=qkv_pairs.device) if triple else \
        torch.empty((B, n, c), dtype=torch.float32, device=qkv_pairs.device)
    scale = scale if scale is not None else d ** -0.5
    base = qkv_pairs.data_ptr()
    _lib.check(sda.lib().sdn_attention_x3_pairs(base, base + 2 * c, base + 4 * c, 3 * c, 3 * c, out.data_ptr(), B, heads, n, n, d, w, w, w, c,
                                                scale, 1 if triple else 0, _lib.stream_ptr()), "sdn_attention_x3_pairs")
    return out


def cross_attention_x3_pairs(q_pairs, kv_pairs, heads, scale=None):
    """q_pairs [B, Nq, 2C] = [hi(q) | lo(q)], kv_pairs [B, Nk, 4C] = [hi(k | v) | lo(k | v)] -> f32 [B, Nq, C]."""
    B, nq, w = q_pairs.shape
    c = w // 2
    nk, d = kv_pairs.shape[1], c // heads
    out = torch.empty((B, nq, c), dtype=torch.float32, device=q_pairs.device)
    scale = scale if scale is not None else d ** -0.5
    kb = kv_pairs.data_ptr()
    _lib.check(sda.lib().sdn_attention_x3_pairs(q_pairs.data_ptr(), kb, kb + 2 * c, c, 2 * c, out.data_ptr(), B, heads, nq, nk, d, 2 * c,
                                                4 * c, 4 * c, c, scale, 0, _lib.stream_ptr()), "sdn_attention_x3_pairs (cross)")
    return out
