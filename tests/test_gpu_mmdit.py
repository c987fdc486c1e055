"""SD-v3 MMDiT (row U7): operator-level checks of the new pieces and whole-network parity vs the CPU oracle on a
small configuration (the full 2 B-parameter model is exercised for shape/finite-ness only: its CPU oracle forward
would take minutes).  Tolerances: fp16 storage -> rel L2 <= 5e-3 vs the fp16-emulating and the fp32 oracle."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import safe_denoiser_amd as sda
from oracle.mmdit import OracleMMDiT
from safe_denoiser_amd import _lib
from safe_denoiser_amd.mmdit import SD3Transformer2DModel
from tests_support import ops

pytestmark = pytest.mark.gpu
H16 = torch.float16


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def test_gemm_rowgate_gelu_tanh_and_broadcast_residual():
    g = torch.Generator().manual_seed(0)
    B, rows, K, N = 3, 48, 256, 256
    a = torch.randn(B * rows, K, generator=g).to(H16); w = (torch.randn(N, K, generator=g) * K ** -0.5).to(H16)
    bias = torch.randn(N, generator=g); gate = torch.randn(B, 2 * N, generator=g); res = torch.randn(B * rows, N, generator=g).to(H16)
    base = a.float() @ w.float().T + bias
    gg = gate.cuda()
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), rowgate=gg[:, N:], residual=res.cuda(), rows_per_batch=rows)
    ref = (base.reshape(B, rows, N) * gate[:, None, N:]).reshape(-1, N) + res.float()
    assert rel_l2(out, ref) <= 5e-4
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), act=3)
    assert rel_l2(out, F.gelu(base, approximate="tanh")) <= 5e-4
    pos = torch.randn(rows, N, generator=g).to(H16)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=pos.cuda(), rows_per_batch=rows, residual_bcast=1)
    assert rel_l2(out, (base.reshape(B, rows, N) + pos.float()[None]).reshape(-1, N)) <= 5e-4


def test_layernorm_mod_patchify_unpatchify():
    g = torch.Generator().manual_seed(1)
    B, rows, Cc = 2, 37, 1536
    x = (torch.randn(B * rows, Cc, generator=g) * 2 + 0.3).to(H16)
    mod = torch.randn(B, 3 * Cc, generator=g) * 0.3
    out = torch.empty_like(x, device="cuda")
    mg, xg = mod.cuda(), x.cuda()
    _lib.check(sda.lib().sdn_layernorm_mod_f16(xg.data_ptr(), B * rows, Cc, 1e-6, mg[:, Cc:].data_ptr(), mg.data_ptr(),
                                               3 * Cc, rows, out.data_ptr(), _lib.stream_ptr()), "ln_mod")
    ref = F.layer_norm(x.float(), (Cc,), eps=1e-6).reshape(B, rows, Cc) * (1 + mod[:, None, Cc:2 * Cc]) + mod[:, None, :Cc]
    assert rel_l2(out, ref.reshape(-1, Cc)) <= 5e-4
    lat = torch.randn(2, 16, 8, 8, generator=g)
    pt = torch.empty(2 * 16, 64, dtype=H16, device="cuda")
    lg = lat.cuda()
    _lib.check(sda.lib().sdn_patchify_f16(lg.data_ptr(), 2, 16, 8, 8, 2, pt.data_ptr(), _lib.stream_ptr()), "patchify")
    ref = F.unfold(lat, kernel_size=2, stride=2).transpose(1, 2).reshape(-1, 64)        # column order (c, py, px)
    assert torch.equal(pt.float().cpu(), ref.half().float())
    tok = torch.randn(2 * 16, 64, generator=g)
    o = torch.empty(2, 16, 8, 8, device="cuda")
    tg = tok.cuda()
    _lib.check(sda.lib().sdn_unpatchify_f32(tg.data_ptr(), 2, 16, 8, 8, 2, o.data_ptr(), _lib.stream_ptr()), "unpatchify")
    ref = torch.einsum("nhwpqc->nchpwq", tok.reshape(2, 4, 4, 2, 2, 16)).reshape(2, 16, 8, 8)
    assert torch.equal(o.cpu(), ref)


@pytest.mark.parametrize("n1,n2", [(64, 13), (1024, 333), (96, 32)])
def test_joint_attention_two_streams(n1, n2):
    g = torch.Generator().manual_seed(2)
    B, Hh, d = 2, 4, 64
    Cc = Hh * d
    qkv1 = torch.randn(B, n1, 3 * Cc, generator=g).to(H16); qkv2 = torch.randn(B, n2, 3 * Cc, generator=g).to(H16)
    cat = torch.cat([qkv1, qkv2], 1).float()
    sp = lambda t: t.reshape(B, n1 + n2, Hh, d).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(cat[..., :Cc]), sp(cat[..., Cc:2 * Cc]), sp(cat[..., 2 * Cc:]))
    ref = ref.transpose(1, 2).reshape(B, n1 + n2, Cc)
    g1, g2 = qkv1.cuda(), qkv2.cuda()
    o1 = torch.empty(B, n1, Cc, dtype=H16, device="cuda"); o2 = torch.empty(B, n2, Cc, dtype=H16, device="cuda")
    s2 = _lib.AttnSegment2(g2.data_ptr(), g2[..., Cc:].data_ptr(), g2[..., 2 * Cc:].data_ptr(), o2.data_ptr(), n1,
                           3 * Cc, 3 * Cc, 3 * Cc, Cc)
    _lib.check(sda.lib().sdn_joint_attention(1, g1.data_ptr(), g1[..., Cc:].data_ptr(), g1[..., 2 * Cc:].data_ptr(),
                                             o1.data_ptr(), C.byref(s2), B, Hh, n1 + n2, d, 3 * Cc, 3 * Cc, 3 * Cc, Cc,
                                             d ** -0.5, _lib.stream_ptr()), "joint_attention")
    assert rel_l2(o1, ref[:, :n1]) <= 1.5e-3 and rel_l2(o2, ref[:, n1:]) <= 1.5e-3


SMALL = dict(sample_size=16, num_layers=3, num_attention_heads=4, joint_attention_dim=128, pooled_projection_dim=64,
             pos_embed_max_size=24)
SMALL_O = dict(sample_size=16, num_layers=3, num_heads=4, joint_dim=128, pooled_dim=64, pos_embed_max_size=24)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 5e-3), (torch.bfloat16, 4e-2)])
def test_small_mmdit_matches_oracle(dtype, tol):
    m = SD3Transformer2DModel(text_len=45, dtype=dtype, **SMALL)
    sd = m.synthetic_state_dict(5)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 16, 16, generator=g); e = torch.randn(2, 45, 128, generator=g); pl = torch.randn(2, 64, generator=g)
    y = m(x.cuda(), timestep=torch.tensor([812.0, 812.0]).cuda(), encoder_hidden_states=e.cuda(),
          pooled_projections=pl.cuda())[0]
    torch.cuda.synchronize()
    assert y.shape == (2, 16, 16, 16) and torch.isfinite(y).all()
    r_em = rel_l2(y, OracleMMDiT(sd, SMALL_O, act_dtype=dtype)(x, 812.0, e, pl))
    r_32 = rel_l2(y, OracleMMDiT(sd, SMALL_O, act_dtype=None)(x, 812.0, e, pl))
    print(f"small mmdit {dtype}: rel L2 vs emulating oracle {r_em:.3e}, vs fp32 oracle {r_32:.3e}")
    assert r_em <= tol and r_32 <= tol


def test_mmdit_batch_above_the_plan_limit_runs_as_row_blocks():
    """SD3Transformer2DModel.max_samples: 170 samples at 512^2 / 42 at 1024^2 is what one plan addresses (31-bit operand offsets on the
    feed-forward hidden state); above it `forward_into` runs consecutive row blocks.  Small configuration with the limit lowered to
    3: 7 samples run as 2 + 2 + 2 + 1 and give the single forward's bits."""
    assert SD3Transformer2DModel(sample_size=64).max_samples() == 170 and SD3Transformer2DModel(sample_size=128).max_samples() == 42
    m = SD3Transformer2DModel(text_len=45, dtype=torch.float16, **SMALL)
    m.load_state_dict(m.synthetic_state_dict(5))
    g = torch.Generator().manual_seed(9)
    x = torch.randn(7, 16, 16, 16, generator=g).cuda(); e = m.prepare_text(torch.randn(7, 45, 128, generator=g).cuda())
    pl = torch.randn(7, 64, generator=g).cuda().half()
    ref, y = torch.empty(7, 16, 16, 16, device="cuda"), torch.empty(7, 16, 16, 16, device="cuda")
    m.forward_into(x, 812.0, e, pl, ref)
    m.max_samples = lambda: 3
    m.forward_into(x, 812.0, e, pl, y)
    assert torch.equal(y, ref)


@pytest.mark.parametrize("side", [64, 128])
def test_full_sd3_medium_plan_runs(side):
    """The 2 B-parameter SD3-medium plan at the reference driver's default 512x512 (latent 64) and at BASELINE config 4's
    1024x1024 (latent 128: 4096 image + 333 text tokens per sample): shapes, finiteness, batch-row independence.
    (Weights are generated on the GPU to keep the test short.)"""
    m = SD3Transformer2DModel(sample_size=side)
    buf = torch.zeros(m.weight_bytes, dtype=torch.uint8, device="cuda")
    gg = torch.Generator(device="cuda").manual_seed(0)
    for p in m.manifest:
        n = p["rows_padded"] * max(p["cols"], 1)
        if p["kind"] == 0:
            t = (torch.rand(n, generator=gg, device="cuda") - 0.5) * 0.2
            buf[p["offset"]:p["offset"] + 4 * n] = t.view(torch.uint8)
        else:
            scale = (3.0 / max(p["cols"], 1)) ** 0.5 * (0.3 if "norm" in p["name"] else 1.0)
            t = ((torch.rand(n, generator=gg, device="cuda") * 2 - 1) * scale).half()
            buf[p["offset"]:p["offset"] + 2 * n] = t.view(torch.uint8)
    m._weights = buf
    x = torch.randn(2, 16, side, side, device="cuda"); e = torch.randn(2, 333, 4096, device="cuda"); pl = torch.randn(2, 2048, device="cuda")
    y = m(x, timestep=900.0, encoder_hidden_states=e, pooled_projections=pl)[0]
    y0 = m(x[:1], timestep=900.0, encoder_hidden_states=e[:1], pooled_projections=pl[:1])[0]
    torch.cuda.synchronize()
    assert y.shape == (2, 16, side, side) and torch.isfinite(y).all()
    assert torch.equal(y0, y[:1])
    total, attn = m.flops(1)
    if side == 64:
        assert abs(total / 1e12 - 2.107) < 0.01


def test_sd3_loop_matches_oracle(tmp_path):
    """Row P4: flow-matching loop with fast_sdv3 repellency re-noise, 2 prompts batched vs the per-prompt oracle on the
    same noise tapes.  fp16 storage + fp16 latents between steps: rel L2 <= 1e-2 on the final latents."""
    from oracle import repellency as orp
    from oracle import schedulers as osch
    from oracle.mmdit import sd3_denoise_one
    from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    m = SD3Transformer2DModel(text_len=45, dtype=torch.float16, **SMALL)
    sd = m.synthetic_state_dict(5)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(7)
    P, steps = 2, 12
    emb = torch.randn(2 * P, 45, 128, generator=g); pooled = torch.randn(2 * P, 64, generator=g)
    refs = orp.channel_normalise(torch.randn(10, 16, 16, 16, generator=g))
    path = str(tmp_path / "pr.pt"); torch.save(refs, path)
    proc = sd3rep.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085,
                                        0.012, n_embed=4, proj_ref_path=path, cache_proj_ref=True, scale=0.03)

    class Tapes:
        def __init__(self):
            gg = torch.Generator().manual_seed(11)
            self.data = [torch.randn(steps + 2, 1, 16, 16, 16, generator=gg) for _ in range(P)]
            self.cur = [0] * P

        def __call__(self, p, shape):
            z = self.data[p][self.cur[p]]
            self.cur[p] += 1
            return z.clone()

    oracle_net = OracleMMDiT(sd, SMALL_O, act_dtype=torch.float16)
    t_o = Tapes()
    ref = []
    for p in range(P):
        pair = torch.stack([emb[p], emb[P + p]]); ppair = torch.stack([pooled[p], pooled[P + p]])
        lat, st = sd3_denoise_one(oracle_net, osch.FlowMatchEuler(), pair, ppair, p, t_o, num_inference_steps=steps,
                                  repel=dict(proj_refs=refs, scale=0.03))
        ref.append(lat)
    ref = torch.cat(ref)
    t_p = Tapes()
    pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
    out = pipe(prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), num_inference_steps=steps,
               repellency_processor=proc, noise_fn=t_p)
    torch.cuda.synchronize()
    assert t_p.cur == t_o.cur and pipe.last_stats["window_steps"] == st["window_steps"] > 0
    errs = [rel_l2(out[p:p + 1], ref[p:p + 1]) for p in range(P)]
    print(f"sd3 loop: window steps {st['window_steps']}, per-prompt rel L2 {['%.2e' % e for e in errs]}")
    assert max(errs) <= 1e-2


def test_sd3_loop_with_safree_text_projection(tmp_path):
    """models/sdv3/safe_denoiser_pipeline.py:1061-1078,1115: the T5-side SAFREE projection feeds the transformer at every
    step.  The pipeline computes it from the caller's first-token states (masked prompt / concept phrases); the result must
    equal passing the finished embeddings, differ from the un-projected run, and match the oracle loop run on that text."""
    from oracle import schedulers as osch
    from oracle.mmdit import sd3_denoise_one
    from safe_denoiser_amd import safree
    from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    m = SD3Transformer2DModel(text_len=45, dtype=torch.float16, **SMALL)
    sd = m.synthetic_state_dict(5)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(17)
    P, steps = 2, 6
    emb = torch.randn(2 * P, 45, 128, generator=g); pooled = torch.randn(2 * P, 64, generator=g)
    negspace = torch.randn(5, 128, generator=g)
    masked = [torch.randn(7, 128, generator=g), torch.randn(11, 128, generator=g)]
    masked[0][2] = negspace[:3].mean(0) * 3                                     # a trigger token
    tape = torch.randn(P, 1, 16, 16, 16, generator=g)
    nf = lambda p, shape: tape[p].clone()
    pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
    kw = dict(prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), num_inference_steps=steps, noise_fn=nf)
    out_a = pipe(masked_embs=[t.cuda() for t in masked], negspace_embs=negspace.cuda(), **kw)
    resc = torch.cat([emb[:P].cuda()] + [safree.prepare_sd3(torch.stack([emb[p], emb[P + p]]).cuda(), masked[p].cuda(), negspace.cuda())
                                         ["rescaled_text_embeddings"][1:2] for p in range(P)])
    out_b = pipe(rescaled_text_embeddings=resc, **kw)
    out_plain = pipe(**kw)
    torch.testing.assert_close(out_a, out_b, rtol=0, atol=0)
    assert float((out_a.float() - out_plain.float()).norm() / out_plain.float().norm()) > 1e-2
    oracle_net = OracleMMDiT(sd, SMALL_O, act_dtype=torch.float16)
    rc = resc.float().cpu()
    ref = torch.cat([sd3_denoise_one(oracle_net, osch.FlowMatchEuler(), torch.stack([rc[p], rc[P + p]]), torch.stack([pooled[p], pooled[P + p]]),
                                     p, nf, num_inference_steps=steps)[0] for p in range(P)])
    r = float((out_a.float().cpu() - ref).norm() / ref.norm())
    print(f"SD3 loop with SAFREE text: rel L2 vs oracle {r:.2e}")
    assert r <= 1e-2


def test_sd3_call_with_prompt_strings_runs_the_references_front_end_orchestration():
    """models/sdv3/safe_denoiser_pipeline.py:862-891,985-1078: `pipe(prompt=[...], negative_prompt=...)`.  The three third-party
    encoders (CLIP x2 + T5) belong to the caller (`text_front_end`); the orchestration between their calls is the reference's: the
    caller's negative prompt is OVERWRITTEN with the 17 joined concept phrases, the masked-prompt / concept states feed the SAFREE
    projection, the projected text goes to the transformer at every step.  With a recording front end the call must equal the
    embeddings-level call on what the front end returned, and the reference's four-tensor form must equal the concatenated one."""
    from safe_denoiser_amd.pipeline_sd3 import SD3_NEGATIVE_PROMPT_SPACE, SD3SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    m = SD3Transformer2DModel(text_len=45, dtype=torch.float16, **SMALL)
    m.load_state_dict(m.synthetic_state_dict(5))
    prompts = ["a lustful portrait in oil", "two cats asleep on a red sofa in the evening sun"]
    P = len(prompts)

    class FrontEnd:                                   # deterministic stand-ins for the encoders' outputs
        def __init__(self):
            self.calls = []

        def _g(self, text):
            import zlib
            return torch.Generator().manual_seed(zlib.crc32(text.encode()))

        def encode_prompt(self, prompt, negative_prompt, **kw):
            self.calls.append(("encode_prompt", list(prompt), list(negative_prompt)))
            pe = torch.stack([torch.randn(45, 128, generator=self._g(p)) for p in prompt])
            ne = torch.stack([torch.randn(45, 128, generator=self._g("neg" + n)) for n in negative_prompt])
            pp = torch.stack([torch.randn(64, generator=self._g("pool" + p)) for p in prompt])
            npp = torch.stack([torch.randn(64, generator=self._g("npool" + n)) for n in negative_prompt])
            return pe.cuda(), ne.cuda(), pp.cuda(), npp.cuda()

        def masked_encode_prompt(self, prompt):
            self.calls.append(("masked", prompt))
            return torch.randn(len(prompt.split()), 128, generator=self._g("m" + prompt)).cuda()

        def encode_negative_prompt_space(self, phrases):
            self.calls.append(("space", list(phrases)))
            return torch.randn(len(phrases), 128, generator=self._g("space")).cuda()

    fe = FrontEnd()
    tape = torch.randn(P, 1, 16, 16, 16, generator=torch.Generator().manual_seed(3))
    nf = lambda p, shape: tape[p].clone()
    pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler(), text_front_end=fe)
    out = pipe(prompt=prompts, negative_prompt="something the reference ignores", num_inference_steps=5, guidance_scale=3.5, noise_fn=nf)
    joined = ", ".join(SD3_NEGATIVE_PROMPT_SPACE)
    assert fe.calls[0] == ("encode_prompt", prompts, [joined] * P)                     # (:996: the caller's negative prompt is overwritten)
    assert [c[0] for c in fe.calls[1:]] == ["masked", "masked", "space"] and fe.calls[-1][1] == SD3_NEGATIVE_PROMPT_SPACE
    # the same call from the embeddings the front end returned, in the reference's four-tensor form and in the concatenated form
    fe2 = FrontEnd()
    pe, ne, pp, npp = fe2.encode_prompt(prompts, [joined] * P)
    masked = [fe2.masked_encode_prompt(p) for p in prompts]
    space = fe2.encode_negative_prompt_space(SD3_NEGATIVE_PROMPT_SPACE)
    plain = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
    kw = dict(num_inference_steps=5, guidance_scale=3.5, noise_fn=nf, masked_embs=masked, negspace_embs=space)
    four = plain(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, negative_pooled_prompt_embeds=npp, **kw)
    cat = plain(prompt_embeds=torch.cat([ne, pe]), pooled_prompt_embeds=torch.cat([npp, pp]), **kw)
    assert torch.equal(out, four) and torch.equal(four, cat)
    with pytest.raises(NotImplementedError):
        plain(prompt=prompts, num_inference_steps=2)                                       # no front end, no embeddings
    with pytest.raises(Exception):
        plain(prompt_embeds=pe, negative_prompt_embeds=ne, pooled_prompt_embeds=pp, num_inference_steps=2)


@pytest.fixture(scope="module")
def sd3_medium():
    """Full SD3-medium (~2 B parameters, synthetic weights seed 3) on the engine + its state_dict, shared by the two full-size tests
    (generating and packing it twice was ~40 s of host time per run of the suite)."""
    m = SD3Transformer2DModel(sample_size=64)
    sd = m.synthetic_state_dict(3)
    m.load_state_dict(sd)
    return m, sd


def test_full_sd3_medium_matches_oracle(sd3_medium):
    """The full SD3-medium configuration (24 joint blocks, 24 heads x 64, ~2 B parameters) at the reference driver's default
    512 x 512 (latent side 64: 1024 image + 333 text tokens), one sample, fp16 storage, against the pure-fp32 oracle.
    Tolerance: 16-bit storage over 24 layers (small model: 6.9e-4; measured here 1.5e-3)."""
    m, sd = sd3_medium
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 16, 64, 64, generator=g); e = torch.randn(1, 333, 4096, generator=g); pl = torch.randn(1, 2048, generator=g)
    y = m(x.cuda(), timestep=812.0, encoder_hidden_states=e.cuda(), pooled_projections=pl.cuda())[0]
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    # the oracle's torch ops evaluated on the GPU (TF32 off), as in the loop test below; its CPU evaluation is what the
    # small-configuration tests above compare with
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    net = OracleMMDiT(sd, None, act_dtype=None, device="cuda")
    r_32 = rel_l2(y, net(x.cuda(), 812.0, e.cuda(), pl.cuda()))
    del net
    torch.cuda.empty_cache()
    print(f"full SD3-medium MMDiT fp16: rel L2 vs the pure-fp32 oracle {r_32:.3e}   (measured once also vs the fp16-emulating oracle: 7.0e-4)")
    assert r_32 <= 5e-3


def test_sd3_call_ends_like_the_reference_with_a_vae():
    """run_nudity_sdv3.py:351-360 reads `pipe(...).images`: with a 16-channel VAE attached and return_latents=False the call
    returns StableDiffusion3PipelineOutput(images=[PIL...]) (safe_denoiser_pipeline.py:1195-1214); output_type="latent" puts
    the latents there; return_dict=False gives a tuple; the engine-side default (return_latents=True) hands back the latents."""
    from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline, StableDiffusion3PipelineOutput
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from safe_denoiser_amd.vae import SD3_VAE_CONFIG, AutoencoderKL
    m = SD3Transformer2DModel(text_len=45, dtype=torch.float16, **SMALL)
    m.load_state_dict(m.synthetic_state_dict(5))
    v = AutoencoderKL(dtype=torch.float16, **dict(SD3_VAE_CONFIG, block_out_channels=(64, 128), layers_per_block=1, sample_size=32))
    v.load_state_dict(v.synthetic_state_dict(6))
    g = torch.Generator().manual_seed(7)
    P = 2
    emb, pooled = torch.randn(2 * P, 45, 128, generator=g).cuda(), torch.randn(2 * P, 64, generator=g).cuda()
    pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler(), vae=v)
    gens = lambda: [torch.Generator(device="cuda").manual_seed(3 + i) for i in range(P)]
    kw = dict(prompt_embeds=emb, pooled_prompt_embeds=pooled, num_inference_steps=4)
    lat = pipe(generator=gens(), **kw)
    assert torch.is_tensor(lat) and tuple(lat.shape) == (P, 16, 16, 16)
    out = pipe(generator=gens(), return_latents=False, **kw)
    assert isinstance(out, StableDiffusion3PipelineOutput) and len(out.images) == P and out.images[0].size == (32, 32)
    as_latent = pipe(generator=gens(), output_type="latent", **kw)
    assert torch.equal(as_latent.images, lat)
    tup = pipe(generator=gens(), return_latents=False, output_type="uint8", return_dict=False, **kw)
    assert isinstance(tup, tuple) and tup[0].dtype == torch.uint8 and tuple(tup[0].shape) == (P, 32, 32, 3)
    import numpy as np
    assert np.array_equal(np.asarray(out.images[1]), tup[0][1].cpu().numpy())


def test_sd3_global_rng_draws_in_one_launch_equal_the_per_prompt_loop(tmp_path):
    """The reference draws its re-noise tensor with `randn_like` on the GLOBAL generator, one prompt after the other
    (safe_denoiser_pipeline.py:1159).  The batched loop takes the P draws of a window step out of ONE launch
    (BatchedNormal.draw_sequence): same latents bit for bit, and the default CUDA generator ends at the same offset."""
    from oracle import repellency as orp
    from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    m = SD3Transformer2DModel(text_len=45, dtype=torch.float16, **SMALL)
    m.load_state_dict(m.synthetic_state_dict(5))
    g = torch.Generator().manual_seed(7)
    P, steps = 3, 12
    emb = torch.randn(2 * P, 45, 128, generator=g).cuda(); pooled = torch.randn(2 * P, 64, generator=g).cuda()
    refs = orp.channel_normalise(torch.randn(10, 16, 16, 16, generator=g))
    path = str(tmp_path / "pr.pt"); torch.save(refs, path)
    proc = sd3rep.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012, n_embed=4,
                                        proj_ref_path=path, cache_proj_ref=True, scale=0.03)
    outs, tails = [], []
    for batched in (True, False, True):
        pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
        pipe.batched_rng = batched
        torch.manual_seed(1234)
        gens = [torch.Generator(device="cuda").manual_seed(50 + p) for p in range(P)]
        outs.append(pipe(prompt_embeds=emb, pooled_prompt_embeds=pooled, num_inference_steps=steps, repellency_processor=proc, generator=gens))
        assert pipe.last_stats["window_steps"] > 0
        tails.append(torch.randn(5, device="cuda"))                          # where the global stream stands afterwards
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(tails[0], tails[1])


def test_full_sd3_medium_10_step_loop_with_fast_sdv3_repellency_matches_oracle(tmp_path, sd3_medium):
    """BASELINE config 4 at full size (VERDICT r3 missing #4): SD3-medium (24 joint blocks, ~2 B parameters) at the reference
    driver's 512 x 512 default, guidance 3.5, 10 flow-Euler steps of which the first 5 (t >= 780) take the repellency re-noise
    path against M = 64 references of [16, 64, 64], 2 prompts batched vs the per-prompt oracle loop on the same tapes (the
    oracle's MMDiT evaluated by torch on the GPU, TF32 off).  fp16 storage + fp16 latents: bound = measured + 25 % against the
    pure-fp32 network (1.9e-3 measured), with the fp16-emulating oracle alongside (1.3e-3)."""
    from oracle import repellency as orp
    from oracle import schedulers as osch
    from oracle.mmdit import sd3_denoise_one
    from safe_denoiser_amd.pipeline_sd3 import SD3SafeDenoiserPipeline
    from safe_denoiser_amd.repellency import repellency_methods_fast_sdv3 as sd3rep
    from safe_denoiser_amd.schedulers import FlowMatchEulerDiscreteScheduler
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    m, sd = sd3_medium
    g = torch.Generator().manual_seed(21)
    P, steps = 2, 10
    emb = torch.randn(2 * P, 333, 4096, generator=g); pooled = torch.randn(2 * P, 2048, generator=g)
    refs = orp.channel_normalise(torch.randn(64, 16, 64, 64, generator=g))
    path = str(tmp_path / "pr.pt"); torch.save(refs, path)
    proc = sd3rep.get_repellency_method("kernel_fast", torch.zeros(1, device="cuda"), None, None, 50, 1000, 0.00085, 0.012, n_embed=4,
                                        proj_ref_path=path, cache_proj_ref=True, scale=0.03)

    class Tapes:
        def __init__(self, dev=None):
            gg = torch.Generator().manual_seed(11)
            self.data = [torch.randn(steps + 2, 1, 16, 64, 64, generator=gg) for _ in range(P)]
            self.cur, self.dev = [0] * P, dev

        def __call__(self, p, shape):
            z = self.data[p][self.cur[p]].clone()
            self.cur[p] += 1
            return z if self.dev is None else z.to(self.dev)

    t_p = Tapes()
    pipe = SD3SafeDenoiserPipeline(m, FlowMatchEulerDiscreteScheduler())
    out = pipe(prompt_embeds=emb.cuda(), pooled_prompt_embeds=pooled.cuda(), num_inference_steps=steps, guidance_scale=3.5,
               repellency_processor=proc, noise_fn=t_p).float().cpu()
    res = {}
    for name, act in (("fp32", None), ("fp16_emulating", torch.float16)):
        net = OracleMMDiT(sd, None, act_dtype=act, device="cuda")
        t_o = Tapes("cuda")
        ref = []
        for p in range(P):
            lat, st = sd3_denoise_one(net, osch.FlowMatchEuler(), torch.stack([emb[p], emb[P + p]]).cuda(),
                                      torch.stack([pooled[p], pooled[P + p]]).cuda(), p, t_o, num_inference_steps=steps, guidance_scale=3.5,
                                      repel=dict(proj_refs=refs.cuda(), scale=0.03))
            ref.append(lat.cpu())
        assert t_p.cur == t_o.cur and pipe.last_stats["window_steps"] == st["window_steps"] > 0
        res[name] = [rel_l2(out[p:p + 1], ref[p]) for p in range(P)]
        del net
        torch.cuda.empty_cache()
    print(f"full SD3-medium, 10-step loop with fast_sdv3 repellency ({st['window_steps']} window steps): rel L2 vs the pure-fp32 oracle "
          f"{['%.2e' % e for e in res['fp32']]}, vs the fp16-emulating oracle {['%.2e' % e for e in res['fp16_emulating']]}")
    # measured on MI355X (round 4): 1.86e-3 / 1.85e-3 vs the pure-fp32 network, 1.28e-3 / 1.30e-3 vs the fp16-emulating one; + 25 %
    assert max(res["fp32"]) <= 2.4e-3 and max(res["fp16_emulating"]) <= 1.65e-3
