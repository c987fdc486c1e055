"""SAFREE wired into the call surface: `pipe(prompt, negative_prompt_space=, safree_dict=)` computes the
masked / concept-space embeddings with the engine's CLIPTextModel, projects the text (...threshold_time.py:186-349,458-486)
and runs the 3-branch loop -- checked stage by stage so that a discrete decision (trigger tokens, f_beta's round) cannot
hide behind another stage's tolerance:
  1. the engine's CLIP outputs vs the oracle text encoder on the same token ids (16-bit storage bound);
  2. the projection on the engine's own embeddings vs the independent numpy restatement (fp32 pinverse bound);
  3. the loop fed with the engine's projected text vs oracle.denoise_one(lra=True, text_safe=...) on the same tapes.
Also: SAFREE helpers on CUDA tensors vs the goldens captured from the reference's own functions."""
import os

import numpy as np
import pytest
import torch

from oracle import pipeline as opipe
from oracle import repellency as orp
from oracle import safree as osf
from oracle import schedulers as osch
from oracle.clip import OracleCLIPText
from oracle.unet import OracleUNet
from safe_denoiser_amd import safree
from safe_denoiser_amd.clip import CLIPTextModel
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
from safe_denoiser_amd.repellency import repellency_methods_threshold as thr
from safe_denoiser_amd.schedulers import DDPMScheduler
from safe_denoiser_amd.unet import UNet2DConditionModel
from tests.test_gpu_pipeline import LOOP_BOUND, SMALL, SMALL_O, Tapes, make_proc, rel_l2
from tests_support.fake_tokenizer import FakeCLIPTokenizer

pytestmark = pytest.mark.gpu

CLIP_CFG = dict(vocab_size=1000, hidden_size=768, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=12,
                max_position_embeddings=77)
NEG_SPACE = ["Sexual Acts", "Content Meant to Arouse Sexual Excitement", "Nudity", "Pornography", "Erotic Art", "Lustful",
             "Seductive", "Orgasmic", "Libido", "Kinky", "Sexual Orientation", "Sexual Attraction", "Sexual Intercourse",
             "Sexual Pleasure", "Sexual Fantasy", "Carnal Desires", "Sexual Gratification"]       # run_nudity.py:353-358
PROMPTS = ["a painting of a woman standing near the sea , lustful mood",
           "portrait photo of an old fisherman , dramatic light , 85mm"]
STEPS = 10


def test_safree_helpers_on_cuda_match_reference_goldens():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "safree_golden.npz"))
    for name in [str(n) for n in z["__cases__"]]:
        ie, neg, p_emb = (torch.from_numpy(z[f"{name}/{k}"]).float().cuda() for k in ("ie", "neg", "p_emb"))
        P_c, P_m = safree.projection_matrix(neg.T), safree.projection_matrix(p_emb.T)
        assert P_c.is_cuda
        resc, n_removed = safree.safree_projection(ie, p_emb, P_m, P_c, alpha=float(z[f"{name}/alpha"]), max_length=77)
        assert n_removed == int(z[f"{name}/n_removed"]), name
        np.testing.assert_allclose(resc.cpu().numpy(), z[f"{name}/rescaled"], atol=3e-3, rtol=3e-3)
        ort = safree.projection_and_orthogonal(ie, P_m, P_c)
        np.testing.assert_allclose(ort.cpu().numpy(), z[f"{name}/proj_ort"], atol=3e-3, rtol=3e-3)


@pytest.fixture(scope="module")
def stack():
    u = UNet2DConditionModel(text_len=77, **SMALL)
    sd = u.synthetic_state_dict(11)
    u.load_state_dict(sd)
    enc = CLIPTextModel(dtype=torch.float16, **CLIP_CFG)
    csd = enc.synthetic_state_dict(31)
    enc.load_state_dict(csd)
    tok = FakeCLIPTokenizer(vocab_size=CLIP_CFG["vocab_size"])
    refs = orp.channel_normalise(torch.randn(24, 4, 16, 16, generator=torch.Generator().manual_seed(4)))
    return u, sd, enc, csd, tok, refs


def test_prompt_call_with_safree_matches_oracle_stage_by_stage(stack, tmp_path):
    u, sd, enc, csd, tok, refs = stack
    P = len(PROMPTS)
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time", text_encoder=enc, tokenizer=tok)
    sf = dict(safree=True, svf=True, lra=True, alpha=0.01, up_t=10, category="nudity", re_attn_t=[-1, 4])
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)
    shape = (1, 4, 16, 16)
    t_p = Tapes(P, shape, 3 * STEPS + 4, seed=33)
    lat = pipe(PROMPTS, num_inference_steps=STEPS, negative_prompt=", ".join(NEG_SPACE), negative_prompt_space=NEG_SPACE,
               repellency_processor=make_proc(thr, refs, tmp_path, **params), safree_dict=sf, noise_fn=t_p, return_latents=True)
    prep = pipe.last_safree
    assert pipe.last_stats["branches"] == 3 and prep is not None

    # ---- stage 1: text encoder outputs (same ids through the oracle encoder with fp16 storage emulation) ----
    oenc = OracleCLIPText(csd, CLIP_CFG, act_dtype=torch.float16)
    ids, mask = tok(PROMPTS, padding="max_length", max_length=77, truncation=True).input_ids, None
    nids = tok([", ".join(NEG_SPACE)] * P, padding="max_length", max_length=77, truncation=True).input_ids
    E_o = torch.cat([oenc(nids), oenc(ids)])
    E_e = pipe._new_encode_prompt(PROMPTS, ", ".join(NEG_SPACE))[0].float().cpu()
    r1 = rel_l2(E_e, E_o)
    sp = tok(NEG_SPACE, padding="max_length", max_length=77, truncation=True)
    neg_o = oenc(sp.input_ids, sp.attention_mask)[torch.arange(len(NEG_SPACE)), sp.input_ids.argmax(-1)]
    r1n = rel_l2(prep["negspace"], neg_o)
    m_ids = pipe._masked_ids(PROMPTS[0])
    n_real = len(PROMPTS[0].split())
    assert m_ids.shape == (n_real, 77) and all(int(m_ids[i, i + 1]) == 0 for i in range(n_real))
    m_o = oenc(m_ids)[torch.arange(n_real), m_ids.argmax(-1)]
    r1m = rel_l2(pipe._masked_encode_prompt(PROMPTS[0]), m_o)
    print(f"text encoder stages: prompt {r1:.2e}, concept space {r1n:.2e}, masked prompt {r1m:.2e}")
    assert max(r1, r1n, r1m) <= 3e-3

    # ---- stage 2: projection, on the ENGINE's embeddings, vs numpy float64 ----
    am = tok(PROMPTS, padding="max_length", max_length=77, truncation=True).attention_mask
    P_c = osf.proj(prep["negspace"].double().cpu().numpy().T)
    for p in range(P):
        masked = pipe._masked_encode_prompt(PROMPTS[p]).double().cpu().numpy()
        pair = np.stack([E_e[p].double().numpy(), E_e[P + p].double().numpy()])
        want, n_removed = osf.safree(pair, masked, 0.01)(osf.proj(masked.T), P_c)
        got = prep["rescaled_text_embeddings"][P + p].double().cpu().numpy()
        assert prep["n_removed"][p] == n_removed, (p, prep["n_removed"], n_removed)
        np.testing.assert_allclose(got, want[1], atol=2e-3, rtol=2e-3)
        act = am[p].numpy() == 1
        ort = ((np.eye(768) - P_c) @ osf.proj(masked.T) @ pair[1].T).T
        cos = np.sum(ort[act] * pair[1][act], -1) / (np.linalg.norm(ort[act], axis=-1) * np.linalg.norm(pair[1][act], axis=-1))
        beta = 1.0 - float(cos.mean())
        assert abs(beta - prep["beta"][p]) <= 2e-3
        assert prep["beta_adjusted"][p] == safree.f_beta(beta, upperbound_timestep=10, concept_type="nudity")
    np.testing.assert_allclose(prep["rescaled_text_embeddings"][:P].cpu().numpy(), E_e[:P].numpy(), atol=0, rtol=0)

    # ---- stage 3: the 3-branch loop with the engine's projected text, vs the oracle loop on the same tapes ----
    unet = OracleUNet(sd, SMALL_O, act_dtype=torch.bfloat16)
    Es = prep["rescaled_text_embeddings"].float().cpu()
    t_o = Tapes(P, shape, 3 * STEPS + 4, seed=33)
    outs, draws = [], 0
    for p in range(P):
        ba = prep["beta_adjusted"][p]
        o, st = opipe.denoise_one(unet, osch.DDPM(), torch.stack([E_e[p], E_e[P + p]]), p, t_o, num_inference_steps=STEPS,
                                  repel=dict(flavour="threshold", proj_refs=refs, **params), lra=True,
                                  text_safe=torch.stack([Es[p], Es[P + p]]), use_safe_fn=lambda i, ba=ba: i <= ba)
        outs.append(o); draws += st["renoise_draws"]
    errs = [rel_l2(lat[p:p + 1], outs[p]) for p in range(P)]
    print(f"safree prompt call: beta_adjusted {prep['beta_adjusted']}, removed tokens {prep['n_removed']}, "
          f"loop rel L2 {['%.2e' % e for e in errs]}")
    assert t_p.cur == t_o.cur and pipe.last_stats["renoise_draws"] == draws > 0
    assert max(errs) <= LOOP_BOUND["safree_call"]              # measured 2.4e-2 + 25 %


def test_prompt_call_requires_concept_space_and_keeps_plain_path(stack):
    u, sd, enc, csd, tok, refs = stack
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), text_encoder=enc, tokenizer=tok)
    with pytest.raises(Exception):
        pipe(PROMPTS[:1], num_inference_steps=2, safree_dict=dict(safree=True), return_latents=True)
    a = pipe(PROMPTS[:1], num_inference_steps=3, noise_fn=Tapes(1, (1, 4, 16, 16), 16, seed=1), return_latents=True)
    b = pipe(prompt_embeddings=pipe.encode_prompt(PROMPTS[:1]), num_inference_steps=3, noise_fn=Tapes(1, (1, 4, 16, 16), 16, seed=1), return_latents=True)
    torch.testing.assert_close(a, b, rtol=0, atol=0)
    with pytest.raises(ValueError):
        pipe(PROMPTS[:2], negative_prompt=["x"], num_inference_steps=1, return_latents=True)


def test_batched_safree_projection_matches_the_per_prompt_path(stack):
    """safree.prepare_batch (one batched float64 Gram / pseudo-inverse / projector for the whole batch, prompts zero-padded to a
    common token count) and safree.prepare prompt by prompt (fp32 torch.pinverse, as the reference runs it), on the engine's own
    CLIP outputs: the same trigger-token decisions and f_beta step counts as each other AND as the float64 numpy oracle on the
    same embeddings.  The projected text: the batched path sits on the oracle (<= 2e-4); the per-prompt fp32 path carries the
    reference's own pseudo-inverse noise (the masked-prompt Gram matrix is near-singular and rcond 1e-15 inverts whatever
    rounding left in its smallest singular values: measured up to 8e-3 on 0.04 % of the elements) -- bounded at 2e-2."""
    u, sd, enc, csd, tok, refs = stack
    prompts = PROMPTS + ["a cat", "w1 w2 w3 w4 w5 w6 w7 w8 w9 w10 w11 w12 w13 w14 w15 w16 w17 w18", "lustful seductive kinky pose , oil painting"]
    sf = dict(safree=True, svf=True, lra=True, alpha=0.01, up_t=10, category="nudity", re_attn_t=[-1, 4], logger=None)
    outs = {}
    for batched in (True, False):
        pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time", text_encoder=enc, tokenizer=tok)
        pipe.batched_safree = batched
        E, _ids, am = pipe._new_encode_prompt(prompts, None)
        outs[batched] = pipe._safree_prepare(prompts, E, am, NEG_SPACE, sf)
    a, b = outs[True], outs[False]
    assert a["n_removed"] == b["n_removed"] and a["beta_adjusted"] == b["beta_adjusted"], (a["n_removed"], b["n_removed"], a["beta"], b["beta"])
    assert torch.equal(a["token_mask"], b["token_mask"]) and sum(a["n_removed"]) > 0
    np.testing.assert_allclose(np.array(a["beta"]), np.array(b["beta"]), atol=2e-3)
    P = len(prompts)
    Ef = E.double().cpu().numpy()
    neg = a["negspace"].double().cpu().numpy()
    worst = {True: 0.0, False: 0.0}
    for p_ in range(P):
        masked = pipe._masked_encode_prompt(prompts[p_]).double().cpu().numpy()
        o = osf.prepare(np.stack([Ef[p_], Ef[P + p_]]), masked, neg, am[p_].numpy(), alpha=0.01)
        assert o["n_removed"] == a["n_removed"][p_] and o["beta_adjusted"] == a["beta_adjusted"][p_], p_
        assert np.array_equal(o["mask"], a["token_mask"][p_].cpu().numpy()), p_
        for batched in (True, False):
            got = outs[batched]["rescaled_text_embeddings"][P + p_].double().cpu().numpy()
            worst[batched] = max(worst[batched], float(np.abs(got - o["rescaled"][1]).max()))
    print(f"SAFREE projected text vs the float64 oracle, max abs: batched (float64 projector) {worst[True]:.2e}, per prompt (fp32 pinverse) {worst[False]:.2e}")
    assert worst[True] <= 2e-4 and worst[False] <= 2e-2
