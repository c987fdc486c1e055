"""The fp32 precision mode (sdn_unet_config.dtype 2, csrc/sdn_f32.hip): operator level vs torch fp32 on the CPU, then the
SAME launch plan as the 16-bit modes against the PURE-fp32 oracle at full SD-v1.4 size -- one forward, and the north
star's loop tolerance (final latents <= 1e-3 rel L2, identical re-noise draws) over a 10-step DDPM run that covers the
780..1000 repellency window.
Tolerances: f32 products and sums on both sides, different summation order (the MFMA is a k-ordered fmaf chain, torch
blocks its sums): rel L2 <= 2e-5 per operator, <= 1e-4 for the whole network (measured ~1e-6 / ~5e-6)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import pipeline as opipe
from oracle import repellency as orp
from oracle import schedulers as osch
from oracle.unet import OracleUNet
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
from safe_denoiser_amd.schedulers import DDPMScheduler
from safe_denoiser_amd.unet import UNet2DConditionModel, _interleave16
from tests_support import ops

pytestmark = pytest.mark.gpu
TOL = 2e-5
# the bf16x3 contraction mode (sdn_gemm_x3 / sdn_attention_x3: operands split into bf16 hi + lo, 16 mantissa bits, three
# MFMAs per product): same operators on the same f32 storage; per-operator bound 3e-5 (measured ~1e-5: 2^-17 per operand,
# averaged over the contraction), per network 1e-4, loop 1e-3 (the north star)
TOL_X3 = 3e-5


@pytest.fixture(params=["f32", "x3"])
def mode(request):
    ops.X3 = request.param == "x3"
    yield request.param
    ops.X3 = False


def tol(mode):
    return TOL_X3 if mode == "x3" else TOL


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(128, 320, 320), (100, 32, 64), (2, 1280, 320), (77 * 3, 640, 768), (1000, 960, 1280), (300, 2560, 320)])
def test_gemm_f32_epilogues(M, N, K, mode):
    TOL = tol(mode)
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a @ w.T + bias
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda()), ref) <= TOL
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), act=1), F.silu(ref + res)) <= TOL
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), out_kind=1), ref) <= TOL
    if N >= 64:
        nv = N - 28                                                      # ragged column count (conv_out: 4 of 32)
        assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), n_valid=nv), ref[:, :nv]) <= TOL


def test_gemm_f32_identity_asymmetric_and_dual_source_rowbias_nchw(mode):
    TOL = tol(mode)
    K = 128
    w = (torch.arange(160 * K).reshape(160, K) % 251 - 125).float()
    assert torch.equal(ops.gemm(torch.eye(K).cuda(), w.cuda()).cpu(), w.T.contiguous())      # exact: catches a transposed C
    B, hw, N, K1, K2 = 3, 64, 320, 640, 320
    a1, a2 = rnd(B * hw, K1, seed=5), rnd(B * hw, K2, seed=6)
    w = rnd(N, K1 + K2, seed=7, scale=(K1 + K2) ** -0.5)
    rb = rnd(B, 2 * N, seed=8).cuda()
    ref = (torch.cat([a1, a2], 1) @ w.T).reshape(B, hw, N) + rb.cpu()[:, None, N:]
    out = ops.gemm(a1.cuda(), w.cuda(), a2=a2.cuda(), rowbias=rb[:, N:], rows_per_batch=hw)
    assert rel_l2(out.reshape(B, hw, N), ref) <= TOL
    out = ops.gemm(a1.cuda(), w.cuda(), a2=a2.cuda(), rowbias=rb[:, N:], rows_per_batch=hw, out_kind=2, n_valid=N - 4)
    assert rel_l2(out, ref[:, :, :N - 4].permute(0, 2, 1)) <= TOL


def test_gemm_f32_geglu(mode):
    TOL = tol(mode)
    M, C = 200, 320
    x, w, b = rnd(M, C, seed=12), rnd(8 * C, C, seed=13, scale=C ** -0.5), rnd(8 * C, seed=14)
    val, gate = (x @ w.T + b).chunk(2, -1)
    out = ops.gemm(x.cuda(), _interleave16(w).contiguous().cuda(), bias=_interleave16(b).contiguous().cuda(), act=2)
    assert rel_l2(out, val * F.gelu(gate)) <= TOL


@pytest.mark.parametrize("B,H,Cin,Cout,stride,ups,asym", [(2, 16, 320, 320, 1, 0, 0), (1, 16, 640, 320, 2, 0, 0),
                                                          (2, 8, 320, 640, 1, 1, 0), (3, 4, 64, 96, 1, 0, 0),
                                                          (2, 16, 128, 128, 2, 0, 1)])
def test_conv3x3_f32(B, H, Cin, Cout, stride, ups, asym, mode):
    TOL = tol(mode)
    x, w, bias = rnd(B, Cin, H, H, seed=15), rnd(Cout, Cin, 3, 3, seed=16, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=17)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, bias, stride=2) if asym else F.conv2d(xin, w, bias, stride=stride, padding=1)
    Ho = ref.shape[-1]
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wn = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    out = ops.gemm(xn, wn, bias=bias.cuda(), conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=Ho, Wo=Ho, stride=stride, upsample=ups, asym_pad=asym))
    assert rel_l2(out.reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2), ref) <= TOL


@pytest.mark.parametrize("B,cin,H,cout", [(2, 4, 16, 320), (1, 4, 64, 320), (3, 16, 8, 96), (1, 4, 8, 322)])
def test_conv_in_f32(B, cin, H, cout):
    """conv_in of the fp32-storage plans (fp32 NCHW latent -> NHWC f32): the weights-in-LDS kernel (cout % 4 == 0) and the
    one-thread-per-output form behind it, vs F.conv2d."""
    import safe_denoiser_amd as sda
    from safe_denoiser_amd import _lib
    x, w, bias = rnd(B, cin, H, H, seed=41), rnd(cout, cin, 3, 3, seed=42, scale=(9 * cin) ** -0.5), rnd(cout, seed=43)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    xg, wg, bg = x.cuda(), w.permute(0, 2, 3, 1).contiguous().cuda(), bias.cuda()
    out = torch.empty(B, H, H, cout, device="cuda")
    _lib.check(sda.lib().sdn_conv_in_f32(xg.data_ptr(), wg.data_ptr(), bg.data_ptr(), B, cin, H, H, cout, out.data_ptr(), _lib.stream_ptr()), "conv_in f32")
    assert rel_l2(out, ref) <= 2e-6


def test_norms_f32():
    B, hw, c1, c2 = 2, 256, 320, 640
    x, x2 = rnd(B, hw, c1, seed=20) + 3.0, rnd(B, hw, c2, seed=21, scale=0.1) + 50.0       # |mean| >> std: no cancellation allowed
    g, b = rnd(c1 + c2, seed=22), rnd(c1 + c2, seed=23)
    ref = F.group_norm(torch.cat([x, x2], -1).permute(0, 2, 1).double(), 32, g.double(), b.double(), 1e-5).permute(0, 2, 1)
    out = ops.groupnorm(x.cuda(), x2.cuda(), 32, 1e-5, 0, g.cuda(), b.cuda())
    assert rel_l2(out, ref) <= TOL
    out = ops.groupnorm(x.cuda(), None, 32, 1e-6, 1, g[:c1].contiguous().cuda(), b[:c1].contiguous().cuda())
    ref1 = F.silu(F.group_norm(x.permute(0, 2, 1).double(), 32, g[:c1].double(), b[:c1].double(), 1e-6)).permute(0, 2, 1)
    assert rel_l2(out, ref1) <= TOL
    y = rnd(1001, 1280, seed=24) + 2.0
    gl, bl = rnd(1280, seed=25), rnd(1280, seed=26)
    assert rel_l2(ops.layernorm(y.cuda(), gl.cuda(), bl.cuda()), F.layer_norm(y.double(), (1280,), gl.double(), bl.double(), 1e-5)) <= TOL


@pytest.mark.parametrize("nq,nk,d", [(256, 256, 40), (100, 77, 80), (64, 333, 160), (200, 77, 64)])
def test_attention_f32(nq, nk, d, mode):
    TOL = tol(mode)
    B, H = 2, 8
    q, kv = rnd(B, nq, H * d, seed=30), rnd(B, nk, 2 * H * d, seed=31)       # k | v fused along the columns, as in the plan
    sp = lambda t: t.reshape(B, -1, H, d).transpose(1, 2).double()

    def ref_of(q_, kv_):
        return F.scaled_dot_product_attention(sp(q_), sp(kv_[..., :H * d]), sp(kv_[..., H * d:])).transpose(1, 2).reshape(B, nq, H * d)

    def run(q_, kv_):
        g = kv_.cuda()
        return ops.attention(q_.cuda(), g[..., :H * d], g[..., H * d:], H)

    assert rel_l2(run(q, kv), ref_of(q, kv)) <= TOL
    spike = kv.clone(); spike[0, 5, :d] *= 40.0                           # one key dominates one head: the running max moves
    assert rel_l2(run(q, spike), ref_of(q, spike)) <= TOL


SMALL = dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
             layers_per_block=1, attention_head_dim=8, cross_attention_dim=768, sample_size=16)
SMALL_O = dict(block_out_channels=(320, 640), level_has_attn=(True, False), layers_per_block=1, n_heads=8,
               cross_dim=768, sample_size=16)


@pytest.mark.parametrize("rep,precision", [(1, "fp32"), (2, "fp32"), (1, "bf16x3"), (2, "bf16x3")])
def test_small_unet_f32_plan_matches_pure_fp32_oracle(rep, precision):
    u = UNet2DConditionModel(text_len=77, precision=precision, latent_repeat=rep, **SMALL)
    sd = u.synthetic_state_dict(11)
    u.load_state_dict(sd)
    x, E = rnd(2, 4, 16, 16, seed=1), rnd(4, 77, 768, seed=2)
    xin = torch.cat([x, x])
    out = u((x if rep == 2 else xin).cuda(), 801.0, encoder_hidden_states=E.cuda()).sample
    ref = OracleUNet(sd, SMALL_O, act_dtype=None)(xin, 801.0, E)
    r = rel_l2(out, ref)
    print(f"small UNet, {precision} plan (latent_repeat {rep}) vs pure-fp32 oracle: rel L2 {r:.2e}")
    assert r <= 1e-4


def _proc(refs, tmp_path, **params):
    path = str(tmp_path / "pr.pt")
    torch.save(refs, path)
    return thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012,
                                     n_embed=4, proj_ref_path=path, cache_proj_ref=True, **params)


# Bounds of the full-size loop test = measured distance + 25 % for the 16-bit storage modes (VERDICT r2 #1d); the two
# fp32-storage modes get a few times their measured distance, far inside the north star's 1e-3.  Measured on MI355X (this
# file's own record, profiles/round3_parity.json): 10 / 50 steps  fp32 1.25e-5 / 1.27e-5, bf16x3 5.1e-5 / 5.4e-5,
# fp16 4.42e-3 / 4.88e-3, bf16 3.51e-2 / 3.83e-2.
LOOP10_BOUND = {"fp32": 5e-5, "bf16x3": 2e-4, "fp16": 5.6e-3, "bf16": 4.4e-2}
LOOP50_BOUND = {"fp32": 5e-5, "bf16x3": 2e-4, "fp16": 6.1e-3, "bf16": 4.8e-2}


def _unet_of(name, sd):
    kw = dict(fp32=dict(precision="fp32"), bf16x3=dict(precision="bf16x3"), fp16=dict(dtype=torch.float16), bf16=dict(dtype=torch.bfloat16))[name]
    u = UNet2DConditionModel(text_len=77, latent_repeat=2, **kw)
    u.load_state_dict(sd)
    return u


def test_full_sd14_every_mode_10_and_50_step_loops_against_the_fp32_oracle(tmp_path, sd14_full_state_dict):
    """Full SD-v1.4 (859.5 M parameters, 64x64x4 latents), 1 prompt, CFG 7.5, DDPM, noise from a tape, repellency against 64
    channel-normalised references with the gate placed so that it fires.  Final latents of the HIP path in EVERY mode
    (fp32 plan, bf16x3 plan, fp16 and bf16 storage) against the pure-fp32 oracle, with identical re-noise draw counts:
      * num_inference_steps = 10 (t = 901, 801 in the repellency window) and
      * num_inference_steps = 50 (the benchmark's loop; 11 window steps)
        vs the oracle's torch ops evaluated on the GPU (TF32 off).  That evaluation is pinned to the oracle on the CPU by ONE
        full-size forward here (<= 1e-5; rounds 3-4 ran the whole 10-step loop on the CPU as well -- 1.0e-5 between the two,
        profiles/round4_parity.json -- at a minute of host matmuls per run of the suite).
    North star: rel L2 <= 1e-3 -- met by the fp32 and bf16x3 plans; the 16-bit storage modes are bounded at their measured
    distance + 25 % (a single 16-bit rounding of the MFMA operands cannot do better than 3e-3: profiles/round3_precision_
    ablation.md).  The record lands in gpurun_out/round5_parity.json (copied to profiles/)."""
    sd = sd14_full_state_dict
    oracle = OracleUNet(sd, None, act_dtype=None)
    g = torch.Generator().manual_seed(5)
    E = torch.randn(2, 77, 768, generator=g)
    x = torch.randn(1, 4, 64, 64, generator=g)
    refs = orp.channel_normalise(torch.randn(64, 4, 64, 64, generator=g))
    tape = torch.randn(40, 1, 4, 64, 64, generator=g)
    tape50 = torch.randn(140, 1, 4, 64, 64, generator=torch.Generator().manual_seed(6))

    class Tape:
        def __init__(self, t, dev=None):
            self.i, self.t = 0, (t if dev is None else t.to(dev))

        def __call__(self, p, shape):
            z = self.t[self.i].reshape(shape).clone()
            self.i += 1
            return z

    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    ref1_cpu = oracle(torch.cat([x, x]), 901.0, E)              # the oracle on the CPU: one forward
    del oracle
    # the same oracle evaluated on the GPU (plain torch ops; no libsdn kernel): the truth of both loops
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    oracle_g = OracleUNet(sd, None, act_dtype=None, device="cuda")
    ref1 = oracle_g(torch.cat([x, x]).cuda(), 901.0, E.cuda()).cpu()
    r_dev = rel_l2(ref1, ref1_cpu)
    t_o = Tape(tape, "cuda")
    ref10, st10 = opipe.denoise_one(oracle_g, osch.DDPM(), E.cuda(), 0, t_o, num_inference_steps=10,
                                    repel=dict(flavour="threshold", proj_refs=refs.cuda(), **params))
    t_g50 = Tape(tape50, "cuda")
    ref50, st50 = opipe.denoise_one(oracle_g, osch.DDPM(), E.cuda(), 0, t_g50, num_inference_steps=50,
                                    repel=dict(flavour="threshold", proj_refs=refs.cuda(), **params))
    del oracle_g
    torch.cuda.empty_cache()
    print(f"oracle on GPU vs oracle on CPU, one full-size forward: rel L2 {r_dev:.2e}; re-noise draws 10 steps {st10['renoise_draws']}, "
          f"50 steps {st50['renoise_draws']}")
    assert r_dev <= 1e-5 and st10["renoise_draws"] == 2 and st50["renoise_draws"] == 11

    res = {}
    import ctypes as C
    import safe_denoiser_amd as sda
    for name in ("fp32", "bf16x3", "fp16", "bf16"):
        un = _unet_of(name, sd)
        fwd = rel_l2(un(x.cuda(), 901.0, encoder_hidden_states=E.cuda()).sample, ref1)
        if name in ("fp16", "bf16"):
            # the 16-bit plans' fusions that re-round a DERIVED weight (LayerNorm folded into W, the FeedForward-output
            # o proj_out product weight [Wpo W2 | Wpo], the one-launch C = 320 feed-forward) must not cost accuracy against the
            # fp32 truth: the forward with each fusion switched off lands within 10 % of the fused forward's distance
            fus = {}
            for hook in ("sdn_debug_set_ff_fuse", "sdn_debug_set_ln_fold", "sdn_debug_set_ffn_fuse", "sdn_debug_set_gn_fuse"):
                getattr(sda.lib(), hook)(C.c_void_p(un._h.value), 0)
                un._ws = {}
                fus[hook[len("sdn_debug_set_"):] + "_off"] = rel_l2(un(x.cuda(), 901.0, encoder_hidden_states=E.cuda()).sample, ref1)
                getattr(sda.lib(), hook)(C.c_void_p(un._h.value), 1)
                un._ws = {}
            print(f"    {name} forward with one fusion off: " + ", ".join(f"{k} {v:.3e}" for k, v in fus.items()) + f" (all on: {fwd:.3e})")
            for k, v in fus.items():
                assert fwd <= 1.10 * v + 1e-4, (name, k, fwd, v)
        pipe = SafeDenoiserPipeline(un, DDPMScheduler(), variant="threshold_time")
        out = {"forward": fwd}
        if name in ("fp16", "bf16"):
            out["forward_with_one_fusion_off"] = fus
        for steps, tp, ref, st, t_ref in ((10, Tape(tape), ref10, st10, t_o), (50, Tape(tape50), ref50, st50, t_g50)):
            lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=steps, guidance_scale=7.5, noise_fn=tp,
                       repellency_processor=_proc(refs, tmp_path, **params), return_latents=True)
            out[f"loop_{steps}"] = rel_l2(lat, ref)
            assert pipe.last_stats["renoise_draws"] == st["renoise_draws"] and tp.i == t_ref.i, (name, steps)
        res[name] = out
        print(f"full SD-v1.4 vs pure-fp32 oracle, {name:6s}: forward {fwd:.2e}, 10-step loop {out['loop_10']:.2e}, "
              f"50-step loop {out['loop_50']:.2e}")
        del un, pipe
        torch.cuda.empty_cache()
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    json.dump({"what": "full SD-v1.4 (859.5 M parameters, synthetic weights seed 1234), 1 prompt, CFG 7.5, DDPM, tape noise, "
                       "repellency gate firing at every window step: rel L2 of the HIP path's output vs the pure-fp32 oracle",
               "source": "tests/test_gpu_f32.py::test_full_sd14_every_mode_10_and_50_step_loops_against_the_fp32_oracle",
               "modes": res, "oracle_gpu_vs_cpu_one_forward": r_dev, "north_star_bound": 1e-3,
               "bounds_10": LOOP10_BOUND, "bounds_50": LOOP50_BOUND}, open(os.path.join(out_dir, "round5_parity.json"), "w"), indent=1)
    assert res["fp32"]["forward"] <= 1e-4 and res["bf16x3"]["forward"] <= 1e-4
    for name in res:
        assert res[name]["loop_10"] <= LOOP10_BOUND[name], (name, res[name])
        assert res[name]["loop_50"] <= LOOP50_BOUND[name], (name, res[name])


def test_bf16x3_plan_in_row_chunks_gives_the_unchunked_bits():
    """The operand-expansion GEMMs address each operand with 31-bit offsets, so a launch whose triple operand reaches 2 GiB (B = 96
    at the 64 x 64 level: 2.26 GB) is cut into row chunks -- whole samples for a convolution (launch_x3t_gemm).  Rows are
    independent: with the limit lowered to 4 MiB (convs one to three samples at a time, GEMMs 256 ... 2048 rows at a time) the small
    UNet's output must be bit-identical to the unchunked forward."""
    import safe_denoiser_amd as sda
    u = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=1, **SMALL)
    u.load_state_dict(u.synthetic_state_dict(11))
    x, E = rnd(3, 4, 16, 16, seed=1).cuda(), rnd(3, 77, 768, seed=2).cuda()
    ref = u(x, 801.0, encoder_hidden_states=E).sample.clone()
    try:
        sda.lib().sdn_debug_set_x3_chunk_bytes(1 << 22)
        out = u(x, 801.0, encoder_hidden_states=E).sample
    finally:
        sda.lib().sdn_debug_set_x3_chunk_bytes(0)
    assert torch.equal(out, ref)


def test_bf16x3_plan_presplit_self_attention_against_the_first_kernel():
    """The bf16x3 plan's self-attention at d = 40 / 80 reads hi | lo pair rows written by the qkv projection (sdn_attention_x3_pairs);
    sdn_debug_set_x3_pairs(0) restores the f32 qkv tensor + sdn_attention_x3.  Same three-term products either way: the two
    forwards agree to the mode's own accuracy, are NOT bit-identical (the switch really changes the path), and both sit at the
    mode's distance from the pure-fp32 oracle."""
    import ctypes as C
    import safe_denoiser_amd as sda
    u = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=1, **SMALL)
    sd = u.synthetic_state_dict(11)
    u.load_state_dict(sd)
    x, E = rnd(3, 4, 16, 16, seed=1), rnd(3, 77, 768, seed=2)
    new = u(x.cuda(), 801.0, encoder_hidden_states=E.cuda()).sample.clone()
    try:
        sda.lib().sdn_debug_set_x3_pairs(C.c_void_p(u._h.value), 0)
        u._ws = {}
        old = u(x.cuda(), 801.0, encoder_hidden_states=E.cuda()).sample.clone()
    finally:
        sda.lib().sdn_debug_set_x3_pairs(C.c_void_p(u._h.value), 1)
    ref = OracleUNet(sd, SMALL_O, act_dtype=None)(x, 801.0, E)
    r_new, r_old, r_between = rel_l2(new, ref), rel_l2(old, ref), rel_l2(new, old)
    print(f"bf16x3 small UNet vs fp32 oracle: pre-split attention {r_new:.2e}, first kernel {r_old:.2e}; between them {r_between:.2e}")
    assert not torch.equal(new, old)
    assert r_new <= 1e-4 and r_old <= 1e-4 and r_between <= 5e-5
