"""CLIP text encoder on the GPU: libsdn's plan vs the transformers golden case (fp32 truth) and the storage-emulating
oracle.  Tolerances: 16-bit storage of a 2-layer (golden) / 12-layer (full) pre-LN transformer: rel L2 <= 1.5e-2 bf16,
<= 2e-3 fp16; helper kernels: one rounding."""
import pytest
import torch

from oracle.clip import OracleCLIPText
from safe_denoiser_amd import _lib
from safe_denoiser_amd.clip import CLIPTextModel
from tests.test_oracle_clip import load_gold

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.5e-2), (torch.float16, 2e-3)])
def test_engine_matches_transformers_golden_case(dtype, tol):
    sd, cfg, ids, mask, plain, masked = load_gold()
    m = CLIPTextModel(dtype=dtype, **cfg)
    m.load_state_dict({"text_model." + k: v for k, v in sd.items()})          # prefixed keys are accepted
    out = m(ids.cuda())
    assert out[0].shape == plain.shape and out.last_hidden_state.dtype == dtype
    r_plain, r_mask = rel_l2(out[0], plain), rel_l2(m(ids.cuda(), attention_mask=mask.cuda())[0], masked)
    r_emul = rel_l2(out[0], OracleCLIPText(sd, cfg, act_dtype=dtype)(ids))
    print(f"CLIP {dtype}: rel L2 vs transformers {r_plain:.3e} (masked {r_mask:.3e}), vs emulating oracle {r_emul:.3e}")
    assert r_plain <= tol and r_mask <= tol and r_emul <= tol
    # pooled output = hidden state at the EOT (highest id) position
    torch.testing.assert_close(out.pooler_output, out[0][torch.arange(3), ids.argmax(-1).cuda()], rtol=0, atol=0)
    # causality: changing a later token must not change earlier positions
    ids2 = ids.clone(); ids2[:, 40:] = 7
    o2 = m(ids2.cuda())[0]
    torch.testing.assert_close(o2[:, :40], out[0][:, :40], rtol=0, atol=0)
    assert float((o2[:, 40:].float() - out[0][:, 40:].float()).abs().max()) > 1e-2


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16x3", 5e-5)])
def test_fp32_storage_modes_match_transformers_golden_case(precision, tol):
    """The reference loads the text encoder in fp32 (run_nudity.py:277): the plan's fp32-storage modes (dtype 2: f32-input matrix
    cores; dtype 3: bf16x3 split-operand GEMMs) against the transformers golden case -- with and without the key-padding mask --
    to fp32 accuracy, f32 hidden states out, the same causality property."""
    sd, cfg, ids, mask, plain, masked = load_gold()
    m = CLIPTextModel(precision=precision, **cfg)
    m.load_state_dict(sd)
    out = m(ids.cuda())
    assert out.last_hidden_state.dtype == torch.float32 and out[0].shape == plain.shape
    r_plain, r_mask = rel_l2(out[0], plain), rel_l2(m(ids.cuda(), attention_mask=mask.cuda())[0], masked)
    print(f"CLIP {precision}: rel L2 vs transformers {r_plain:.3e} (masked {r_mask:.3e})")
    assert r_plain <= tol and r_mask <= tol
    ids2 = ids.clone(); ids2[:, 40:] = 7
    torch.testing.assert_close(m(ids2.cuda())[0][:, :40], out[0][:, :40], rtol=0, atol=0)
    with pytest.raises(_lib.SdnError):
        CLIPTextModel(precision="fp8", **cfg)


def test_masked_attention_f32_against_torch():
    """sdn_masked_attention_f32 alone: causal, key-padding, and both, on strided q / k / v (the stacked qkv buffer's layout)."""
    import safe_denoiser_amd as sda
    g = torch.Generator().manual_seed(4)
    B, H, n, d = 3, 12, 77, 64
    qkv = torch.randn(B, n, 3 * H * d, generator=g).cuda()
    km = torch.ones(B, n, dtype=torch.int32); km[0, 21:] = 0; km[1, 51:] = 0
    kmg = km.cuda()
    q, k, v = (qkv[:, :, i * H * d:(i + 1) * H * d].reshape(B, n, H, d).transpose(1, 2) for i in range(3))
    for causal, mask in ((1, None), (0, kmg), (1, kmg)):
        bias = torch.zeros(B, 1, n, n, device="cuda")
        if causal:
            bias = bias + torch.full((n, n), float("-inf"), device="cuda").triu_(1)
        if mask is not None:
            bias = bias + torch.where(mask[:, None, None, :] != 0, 0.0, float("-inf"))
        ref = (torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + bias, dim=-1) @ v).transpose(1, 2).reshape(B, n, H * d)
        out = torch.full((B, n, H * d), float("nan"), device="cuda")
        _lib.check(sda.lib().sdn_masked_attention_f32(qkv.data_ptr(), qkv.data_ptr() + 4 * H * d, qkv.data_ptr() + 8 * H * d, out.data_ptr(),
                                                      None if mask is None else mask.data_ptr(), causal, B, H, n, d, 3 * H * d, 3 * H * d,
                                                      3 * H * d, H * d, d ** -0.5, _lib.stream_ptr()), "masked attention f32")
        if not causal:                          # without the causal mask a padded QUERY row still sees the valid keys: compare all rows
            pass
        assert rel_l2(out, ref) <= 2e-6, (causal, mask is not None, rel_l2(out, ref))
    assert sda.lib().sdn_masked_attention_f32(qkv.data_ptr(), qkv.data_ptr(), qkv.data_ptr(), out.data_ptr(), None, 0, B, H, n, d,
                                              3 * H * d, 3 * H * d, 3 * H * d, H * d, 0.125, _lib.stream_ptr()) != 0    # no mask at all: refused


@pytest.mark.parametrize("precision,tol", [("fp32", 3e-5), ("bf16x3", 1e-4)])
def test_full_size_text_encoder_fp32_storage_modes(precision, tol):
    """All 12 layers / 123 M parameters, 5 sequences (a ragged batch), vs the pure-fp32 oracle; the UNet's fp32-storage plans
    take the f32 states as they are."""
    m = CLIPTextModel(precision=precision)
    sd = m.synthetic_state_dict(21)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, 49000, (5, 77), generator=g); ids[:, 0] = 49406
    for b, n in enumerate((9, 30, 77, 3, 50)):
        ids[b, n:] = 49407
    am = (torch.arange(77)[None, :] <= ids.argmax(-1, keepdim=True)).long()
    ref = OracleCLIPText(sd, None, act_dtype=None)
    r = rel_l2(m(ids.cuda())[0], ref(ids))
    rm = rel_l2(m(ids.cuda(), attention_mask=am.cuda())[0], ref(ids, am))
    print(f"full CLIP text encoder, {precision}: rel L2 vs pure-fp32 oracle {r:.3e} (with the key-padding mask {rm:.3e})")
    assert r <= tol and rm <= tol


def test_full_size_text_encoder_matches_oracle():
    m = CLIPTextModel()
    sd = m.synthetic_state_dict(21)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(1, 49000, (2, 77), generator=g); ids[:, 0] = 49406; ids[0, 9:] = 49407; ids[1, 30:] = 49407
    out = m(ids.cuda())[0]
    assert out.shape == (2, 77, 768) and torch.isfinite(out.float()).all()
    ref = OracleCLIPText(sd, None, act_dtype=torch.bfloat16)(ids)
    r = rel_l2(out, ref)
    print(f"full CLIP text encoder: rel L2 vs bf16-emulating oracle {r:.3e}")
    assert r <= 1.5e-2
    with pytest.raises(_lib.SdnError):
        m(ids[:, :50].cuda())
    # the UNet consumes it directly (16-bit [B, 77, 768])
    from safe_denoiser_amd.unet import UNet2DConditionModel
    u = UNet2DConditionModel(text_len=77, block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                             layers_per_block=1, attention_head_dim=8, cross_attention_dim=768, sample_size=16)
    u.load_state_dict(u.synthetic_state_dict(7))
    y = u(torch.randn(2, 4, 16, 16).cuda(), 500.0, encoder_hidden_states=out).sample
    assert torch.isfinite(y).all()


def test_pipeline_accepts_prompt_strings_with_a_text_encoder_and_tokenizer():
    """The call surface of the reference's pipelines with strings: tokenizer (the caller's) -> engine text encoder ->
    loop.  Same embeddings passed explicitly give the same latents bit for bit."""
    from types import SimpleNamespace

    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import DDIMScheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    sd, cfg, ids, mask, plain, masked = load_gold()

    def tokenizer(texts, padding=None, max_length=None, truncation=None, return_tensors=None):     # stand-in for CLIPTokenizer
        assert padding == "max_length" and max_length == 77 and truncation
        rows = []
        for t in texts:
            body = [1 + (ord(ch) % 400) for ch in t][:75]
            rows.append([510] + body + [511] * (76 - len(body)))
        return SimpleNamespace(input_ids=torch.tensor(rows))

    te = CLIPTextModel(**cfg)
    te.load_state_dict(sd)
    u = UNet2DConditionModel(text_len=77, latent_repeat=2, block_out_channels=(320, 640),
                             down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), layers_per_block=1,
                             attention_head_dim=8, cross_attention_dim=128, sample_size=16)
    u.load_state_dict(u.synthetic_state_dict(7))
    pipe = SafeDenoiserPipeline(u, DDIMScheduler(), text_encoder=te, tokenizer=tokenizer)
    gens = lambda: [torch.Generator(device="cuda").manual_seed(3 + i) for i in range(2)]
    prompts = ["a photo of a cat", "an oil painting of a ship"]
    lat = pipe(prompt=prompts, num_inference_steps=3, generator=gens(), return_latents=True)
    E = pipe.encode_prompt(prompts)
    assert E.shape == (4, 77, 128)
    torch.testing.assert_close(pipe(prompt_embeddings=E, num_inference_steps=3, generator=gens(), return_latents=True), lat, rtol=0, atol=0)
    with pytest.raises(NotImplementedError):
        SafeDenoiserPipeline(u, DDIMScheduler())(prompt="x", num_inference_steps=1, return_latents=True)
