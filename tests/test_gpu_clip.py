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
