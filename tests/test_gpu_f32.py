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


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(128, 320, 320), (100, 32, 64), (2, 1280, 320), (77 * 3, 640, 768), (1000, 960, 1280)])
def test_gemm_f32_epilogues(M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    ref = a @ w.T + bias
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda()), ref) <= TOL
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), act=1), F.silu(ref + res)) <= TOL
    assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), out_kind=1), ref) <= TOL
    if N >= 64:
        nv = N - 28                                                      # ragged column count (conv_out: 4 of 32)
        assert rel_l2(ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), n_valid=nv), ref[:, :nv]) <= TOL


def test_gemm_f32_identity_asymmetric_and_dual_source_rowbias_nchw():
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


def test_gemm_f32_geglu():
    M, C = 200, 320
    x, w, b = rnd(M, C, seed=12), rnd(8 * C, C, seed=13, scale=C ** -0.5), rnd(8 * C, seed=14)
    val, gate = (x @ w.T + b).chunk(2, -1)
    out = ops.gemm(x.cuda(), _interleave16(w).contiguous().cuda(), bias=_interleave16(b).contiguous().cuda(), act=2)
    assert rel_l2(out, val * F.gelu(gate)) <= TOL


@pytest.mark.parametrize("B,H,Cin,Cout,stride,ups,asym", [(2, 16, 320, 320, 1, 0, 0), (1, 16, 640, 320, 2, 0, 0),
                                                          (2, 8, 320, 640, 1, 1, 0), (3, 4, 64, 96, 1, 0, 0),
                                                          (2, 16, 128, 128, 2, 0, 1)])
def test_conv3x3_f32(B, H, Cin, Cout, stride, ups, asym):
    x, w, bias = rnd(B, Cin, H, H, seed=15), rnd(Cout, Cin, 3, 3, seed=16, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=17)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, bias, stride=2) if asym else F.conv2d(xin, w, bias, stride=stride, padding=1)
    Ho = ref.shape[-1]
    xn = x.permute(0, 2, 3, 1).contiguous().cuda()
    wn = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().cuda()
    out = ops.gemm(xn, wn, bias=bias.cuda(), conv=dict(Hs=H, Ws=H, Cin=Cin, Ho=Ho, Wo=Ho, stride=stride, upsample=ups, asym_pad=asym))
    assert rel_l2(out.reshape(B, Ho, Ho, Cout).permute(0, 3, 1, 2), ref) <= TOL


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
def test_attention_f32(nq, nk, d):
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


@pytest.mark.parametrize("rep", [1, 2])
def test_small_unet_f32_plan_matches_pure_fp32_oracle(rep):
    u = UNet2DConditionModel(text_len=77, dtype=torch.float32, latent_repeat=rep, **SMALL)
    sd = u.synthetic_state_dict(11)
    u.load_state_dict(sd)
    x, E = rnd(2, 4, 16, 16, seed=1), rnd(4, 77, 768, seed=2)
    xin = torch.cat([x, x])
    out = u((x if rep == 2 else xin).cuda(), 801.0, encoder_hidden_states=E.cuda()).sample
    ref = OracleUNet(sd, SMALL_O, act_dtype=None)(xin, 801.0, E)
    r = rel_l2(out, ref)
    print(f"small UNet, fp32 plan (latent_repeat {rep}) vs pure-fp32 oracle: rel L2 {r:.2e}")
    assert r <= 1e-4


def _proc(refs, tmp_path, **params):
    path = str(tmp_path / "pr.pt")
    torch.save(refs, path)
    return thr.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012,
                                     n_embed=4, proj_ref_path=path, cache_proj_ref=True, **params)


def test_full_sd14_fp32_plan_forward_and_10_step_loop_meet_the_north_star_tolerance(tmp_path):
    """Full SD-v1.4 (859.5 M parameters, 64x64x4 latents), 1 prompt, CFG 7.5, DDPM, num_inference_steps=10 (t = 901, 801
    in the repellency window), noise from a tape, repellency against 64 channel-normalised references with the gate
    placed so that it fires: final latents of the HIP path vs the pure-fp32 CPU oracle, rel L2 <= 1e-3 (north star) with
    identical re-noise draw counts; the 16-bit modes' distance from the same truth is printed beside it."""
    u = UNet2DConditionModel(text_len=77, dtype=torch.float32, latent_repeat=2)
    sd = u.synthetic_state_dict(1234)
    u.load_state_dict(sd)
    oracle = OracleUNet(sd, None, act_dtype=None)
    g = torch.Generator().manual_seed(5)
    E = torch.randn(2, 77, 768, generator=g)
    x = torch.randn(1, 4, 64, 64, generator=g)
    ref1 = oracle(torch.cat([x, x]), 901.0, E)
    out1 = u(x.cuda(), 901.0, encoder_hidden_states=E.cuda()).sample
    r_fwd = rel_l2(out1, ref1)
    print(f"full SD-v1.4 UNet, fp32 plan vs pure-fp32 oracle: rel L2 {r_fwd:.2e}")
    assert r_fwd <= 1e-4

    refs = orp.channel_normalise(torch.randn(64, 4, 64, 64, generator=g))
    tape = torch.randn(40, 1, 4, 64, 64, generator=g)

    class Tape:
        def __init__(self):
            self.i = 0

        def __call__(self, p, shape):
            z = tape[self.i].reshape(shape).clone()
            self.i += 1
            return z

    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    steps = 10
    t_o = Tape()
    ref, st = opipe.denoise_one(oracle, osch.DDPM(), E, 0, t_o, num_inference_steps=steps,
                                repel=dict(flavour="threshold", proj_refs=refs, **params))
    res = {}
    import os
    # the 16-bit arms (two more 860 M-parameter packs and loops) only when asked: SDN_PARITY_FULL=1 refreshes the record
    # profiles/round2_parity.json; the default run asserts the fp32 plan alone
    arms = (("fp32", torch.float32), ("fp16", torch.float16), ("bf16", torch.bfloat16)) if os.environ.get("SDN_PARITY_FULL") else \
        (("fp32", torch.float32),)
    for name, dt in arms:
        un = u if dt == torch.float32 else UNet2DConditionModel(text_len=77, dtype=dt, latent_repeat=2)
        if dt != torch.float32:
            un.load_state_dict(sd)
        t_p = Tape()
        pipe = SafeDenoiserPipeline(un, DDPMScheduler(), variant="threshold_time")
        lat = pipe(prompt_embeddings=E.cuda(), num_inference_steps=steps, guidance_scale=7.5, noise_fn=t_p,
                   repellency_processor=_proc(refs, tmp_path, **params), return_latents=True)
        res[name] = (rel_l2(lat, ref), pipe.last_stats["renoise_draws"], t_p.i)
        del un
    print(f"full SD-v1.4 10-step loop vs pure-fp32 oracle (re-noise draws {st['renoise_draws']}): " +
          ", ".join(f"{k} {v[0]:.2e}" for k, v in res.items()))
    import json
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    if len(arms) == 3:
        json.dump({"what": "full SD-v1.4 (859.5 M parameters), 1 prompt, CFG 7.5, DDPM, 10 steps (2 in the repellency window, gate "
                       "fires both times), tape noise: rel L2 of the HIP path's final latents vs the pure-fp32 CPU oracle",
               "source": "tests/test_gpu_f32.py::test_full_sd14_fp32_plan_forward_and_10_step_loop_meet_the_north_star_tolerance",
               "unet_forward_fp32_plan": r_fwd, "loop_10_steps": {k: v[0] for k, v in res.items()},
               "renoise_draws": {k: v[1] for k, v in res.items()}, "oracle_renoise_draws": st["renoise_draws"],
               "north_star_bound": 1e-3}, open(os.path.join(out_dir, "round2_parity.json"), "w"), indent=1)
    assert res["fp32"][1] == st["renoise_draws"] == 2 and res["fp32"][2] == t_o.i
    assert res["fp32"][0] <= 1e-3                                        # the north-star bound
    if len(arms) == 3:
        assert res["fp16"][0] <= 3e-2 and res["bf16"][0] <= 2e-1          # 16-bit storage: reported, loosely bounded
