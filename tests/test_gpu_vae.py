"""VAE decoder row on the GPU: libsdn's decoder plan and its helper kernels vs the CPU oracle.

Tolerances: the helper kernels are fp32 (<= 1e-5) or one 16-bit rounding of the output; the whole decoder follows the
UNet's reasoning (tests/test_gpu_unet.py): 16-bit STORAGE of ~45 normalised layers decorrelates rounding, so the bf16
engine is compared with the bf16-emulating oracle at rel L2 <= 2.5e-2 and the fp16 engine at <= 3e-3; the uint8 image
may differ by one code value on a small fraction of pixels.
"""
import pytest
import torch

from oracle.vae import OracleVAEDecoder
from safe_denoiser_amd import _lib
from safe_denoiser_amd.vae import AutoencoderKL

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def test_latent_mix_softmax_transpose_postprocess_match_torch():
    lib = _lib.lib()
    g = torch.Generator().manual_seed(0)
    # 1x1 conv on an fp32 NCHW latent with the input scale folded in
    z = torch.randn(3, 4, 8, 8, generator=g)
    w, b = torch.randn(4, 4, generator=g), torch.randn(4, generator=g)
    zd, wd, bd = z.cuda(), w.cuda(), b.cuda()
    out = torch.empty_like(zd)
    _lib.check(lib.sdn_latent_mix(zd.data_ptr(), wd.data_ptr(), bd.data_ptr(), 3, 4, 64, 5.49, out.data_ptr(), _lib.stream_ptr()), "mix")
    ref = torch.nn.functional.conv2d(z * 5.49, w[:, :, None, None], b)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    # row softmax, fp32 -> 16 bit, ragged row count, full 4096 width and a narrow one
    for n, rows in ((4096, 5), (64, 64)):
        s = (torch.randn(rows, n, generator=g) * 30).cuda()
        for dt, code, tol in ((torch.bfloat16, 0, 4e-3), (torch.float16, 1, 5e-4)):
            p = torch.empty(rows, n, dtype=dt, device="cuda")
            _lib.check(lib.sdn_softmax_rows(code, s.data_ptr(), n, rows, n, 0.0442, p.data_ptr(), n, _lib.stream_ptr()), "softmax")
            ref = torch.softmax(s.cpu() * 0.0442, dim=-1)
            assert rel_l2(p, ref) <= tol
            torch.testing.assert_close(p.float().sum(-1).cpu(), torch.ones(rows), rtol=0, atol=2e-2)
    # transpose of a strided view
    m = torch.randn(100, 3 * 72, generator=g).bfloat16().cuda()
    t = torch.empty(72, 100, dtype=torch.bfloat16, device="cuda")
    view = m[:, 72:144]
    _lib.check(lib.sdn_transpose16(view.data_ptr(), 100, 72, 3 * 72, t.data_ptr(), 100, _lib.stream_ptr()), "transpose")
    torch.testing.assert_close(t.cpu(), view.cpu().t().contiguous(), rtol=0, atol=0)
    # image post-processing: exact, including the half-to-even rounding and the clamp
    img = torch.randn(2, 3, 16, 24, generator=g) * 1.5
    img[0, 0, 0, :5] = torch.tensor([-1.0, 1.0, 2.0 * (0.5 / 255) - 1.0, 2.0 * (1.5 / 255) - 1.0, float("inf")])
    v = AutoencoderKL.__new__(AutoencoderKL)
    f = AutoencoderKL.postprocess(v, img.cuda()).cpu()
    u = AutoencoderKL.postprocess(v, img.cuda(), uint8=True).cpu()
    ref01 = (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous()
    torch.testing.assert_close(f, ref01, rtol=0, atol=1e-7)
    assert int((u.int() - (ref01 * 255).round().int()).abs().max()) == 0


SMALL = dict(block_out_channels=(64, 128), layers_per_block=1, sample_size=16)
SMALL_O = dict(block_out_channels=(64, 128), layers_per_block=1)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2.5e-2), (torch.float16, 3e-3)])
def test_small_decoder_matches_oracle(dtype, tol):
    v = AutoencoderKL(dtype=dtype, **SMALL)
    sd = v.synthetic_state_dict(3)
    v.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(3, 4, 8, 8, generator=g)
    img = v.decode(z.cuda(), latent_scale=1.0 / 0.18215).sample
    assert img.shape == (3, 3, 16, 16) and torch.isfinite(img).all()
    ref_q = OracleVAEDecoder(sd, SMALL_O, act_dtype=dtype).decode(z, 1.0 / 0.18215)
    ref_32 = OracleVAEDecoder(sd, SMALL_O, act_dtype=None).decode(z, 1.0 / 0.18215)
    r1, r2 = rel_l2(img, ref_q), rel_l2(img, ref_32)
    print(f"small VAE decoder {dtype}: rel L2 vs emulating oracle {r1:.3e}, vs fp32 oracle {r2:.3e}")
    assert r1 <= tol and r2 <= tol
    # batch rows are independent
    torch.testing.assert_close(v.decode(z[1:2].cuda(), latent_scale=1.0 / 0.18215).sample, img[1:2], rtol=0, atol=0)


def test_full_sd14_decoder_matches_oracle_and_pipeline_tail():
    v = AutoencoderKL()
    sd = v.synthetic_state_dict(11)
    v.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 4, 64, 64, generator=g) * 0.18215 * 3.0          # decode_latents divides by the scaling factor
    img = v.decode(lat.cuda(), latent_scale=1.0 / 0.18215).sample
    assert img.shape == (1, 3, 512, 512) and torch.isfinite(img).all()
    o = OracleVAEDecoder(sd, None, act_dtype=torch.bfloat16)
    ref = o.decode(lat, 1.0 / 0.18215)
    r = rel_l2(img, ref)
    print(f"full SD-v1.4 VAE decoder: rel L2 vs bf16-emulating oracle {r:.3e}; |y| rms {float(ref.pow(2).mean().sqrt()):.3f}")
    assert r <= 2.5e-2
    # the pipelines' tail: decode_latents (NHWC fp32 numpy in [0,1]) and numpy_to_pil's uint8
    im01 = v.decode_latents(lat.cuda())
    assert im01.shape == (1, 512, 512, 3) and im01.dtype.name == "float32"
    ref01 = (ref / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
    assert float((torch.from_numpy(im01) - ref01).abs().mean()) <= 5e-3
    u8 = v.decode_latents_uint8(lat.cuda()).cpu()
    d = (u8.int() - o.to_uint8(ref01).int()).abs()
    print(f"uint8 image: {float((d > 0).float().mean()) * 100:.1f} % of values differ, max {int(d.max())} code values")
    assert float(d.float().mean()) <= 1.5


def test_decode_chunks_large_batches_and_rejects_bad_shapes():
    v = AutoencoderKL(**SMALL)
    v.load_state_dict(v.synthetic_state_dict(3))
    z = torch.randn(19, 4, 8, 8).cuda()                       # > MAX_CHUNK: three plan invocations
    img = v.decode(z).sample
    torch.testing.assert_close(v.decode(z[17:18]).sample, img[17:18], rtol=0, atol=0)
    with pytest.raises(_lib.SdnError):
        v.decode(torch.randn(1, 4, 16, 16).cuda())
    with pytest.raises(_lib.SdnUnavailable):
        v.decode(torch.randn(1, 4, 8, 8))                      # host tensor: no CPU fallback


def test_pipeline_ends_like_the_reference_with_images():
    """Steps 8-10 of the reference's __call__ (...threshold_time.py:588-596) behind the same loop: with a VAE attached,
    return_latents=False yields decode_latents' NHWC [0,1] array / numpy_to_pil's images of the SAME latents."""
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import DDPMScheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    u = UNet2DConditionModel(text_len=77, block_out_channels=(320, 640),
                             down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), layers_per_block=1,
                             attention_head_dim=8, cross_attention_dim=768, sample_size=16)
    u.load_state_dict(u.synthetic_state_dict(11))
    v = AutoencoderKL(block_out_channels=(64, 128), layers_per_block=1, sample_size=32)     # latent side 16
    sd = v.synthetic_state_dict(4)
    v.load_state_dict(sd)
    E = torch.randn(4, 77, 768, generator=torch.Generator().manual_seed(2)).cuda()
    gens = lambda: [torch.Generator(device="cuda").manual_seed(7 + i) for i in range(2)]
    pipe = SafeDenoiserPipeline(u, DDPMScheduler(), variant="threshold_time", vae=v)
    lat = pipe(prompt_embeddings=E, num_inference_steps=4, generator=gens(), return_latents=True)
    im_np = pipe(prompt_embeddings=E, num_inference_steps=4, generator=gens(), return_latents=False, output_type="np")
    im_u8 = pipe(prompt_embeddings=E, num_inference_steps=4, generator=gens(), return_latents=False, output_type="uint8")
    im_pil = pipe(prompt_embeddings=E, num_inference_steps=4, generator=gens(), return_latents=False)
    assert im_np.shape == (2, 32, 32, 3) and im_np.dtype.name == "float32"
    import numpy as np
    assert np.array_equal(im_np, v.decode_latents(lat))                              # same latents, same decoder
    assert np.array_equal(im_u8.cpu().numpy(), (im_np * 255).round().astype("uint8"))
    assert len(im_pil) == 2 and im_pil[0].size == (32, 32) and np.array_equal(np.asarray(im_pil[1]), im_u8[1].cpu().numpy())
    ref = OracleVAEDecoder(sd, dict(block_out_channels=(64, 128), layers_per_block=1), act_dtype=torch.bfloat16).decode_latents(lat.cpu())
    assert float((torch.from_numpy(im_np) - ref).abs().mean()) <= 5e-3
    with pytest.raises(NotImplementedError):
        SafeDenoiserPipeline(u, DDPMScheduler())(prompt_embeddings=E, num_inference_steps=2, return_latents=False)


# ------------------------------------------------------------------------------------------ encoder half / proj_ref builder
from oracle.vae import OracleVAEEncoder  # noqa: E402


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 2.5e-2), (torch.float16, 3e-3)])
def test_small_encoder_matches_oracle(dtype, tol):
    v = AutoencoderKL(dtype=dtype, **SMALL)
    sd = v.synthetic_state_dict(3, with_encoder=True)
    v.load_state_dict(sd)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 3, 16, 16, generator=g).clamp(-1, 1)
    dist = v.encode(x.cuda()).latent_dist
    assert dist.parameters.shape == (3, 8, 8, 8)
    ref_q = OracleVAEEncoder(sd, SMALL_O, act_dtype=dtype).encode(x)
    ref_32 = OracleVAEEncoder(sd, SMALL_O, act_dtype=None).encode(x)
    r1, r2 = rel_l2(dist.parameters, ref_q), rel_l2(dist.parameters, ref_32)
    print(f"small VAE encoder {dtype}: moments rel L2 vs emulating oracle {r1:.3e}, vs fp32 oracle {r2:.3e}")
    assert r1 <= tol and r2 <= tol
    # latent_dist: mean / clamped logvar / std views, mode, and sample = mean + std * randn(generator) * scale
    torch.testing.assert_close(dist.mode(0.18215), dist.mean * 0.18215, rtol=1e-6, atol=1e-7)
    gen = torch.Generator(device="cuda").manual_seed(9)
    z = dist.sample(gen, scale=0.18215)
    noise = torch.randn(dist.mean.shape, generator=torch.Generator(device="cuda").manual_seed(9), device="cuda")
    torch.testing.assert_close(z, (dist.mean + dist.std * noise) * 0.18215, rtol=1e-5, atol=1e-6)


def test_full_sd14_encoder_matches_oracle():
    v = AutoencoderKL()
    sd = v.synthetic_state_dict(11, with_encoder=True)
    v.load_state_dict(sd)
    x = (torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(6)) * 2 - 1)
    mom = v.encode(x.cuda()).latent_dist.parameters
    assert mom.shape == (1, 8, 64, 64) and torch.isfinite(mom).all()
    ref = OracleVAEEncoder(sd, None, act_dtype=torch.bfloat16).encode(x)
    r = rel_l2(mom, ref)
    print(f"full SD-v1.4 VAE encoder: moments rel L2 vs bf16-emulating oracle {r:.3e}")
    assert r <= 2.5e-2


def test_proj_ref_builder_runs_on_the_engine(tmp_path):
    """RepellencyMethod.project / set_proj_ref (repellency_methods_threshold.py:54-106) with the engine's own embed_fn:
    chunks of n_embed, channel normalisation, the saved cache -- and chunking does not change the result."""
    from oracle import repellency as orp
    from safe_denoiser_amd.repellency import repellency_methods_fast as thr
    v = AutoencoderKL(**SMALL)
    sd = v.synthetic_state_dict(3, with_encoder=True)
    v.load_state_dict(sd)
    imgs = (torch.rand(5, 3, 16, 16, generator=torch.Generator().manual_seed(8)) * 2 - 1).cuda()
    embed = lambda x: v.encode(x).latent_dist.mode(scale=v.config.scaling_factor)       # noise-free: comparable across chunkings
    outs = []
    for n_embed in (2, 16):
        path = str(tmp_path / f"refs_{n_embed}.pt")
        proc = thr.get_repellency_method("kernel_fast", imgs, embed, None, 50, 1000, 0.00085, 0.012, n_embed=n_embed,
                                         proj_ref_path=path, cache_proj_ref=False, scale=0.03, sigma=1.0, epsilon=1e-8)
        outs.append(proc.proj_refs.float().cpu())
        torch.testing.assert_close(torch.load(path).float(), outs[-1])
    torch.testing.assert_close(outs[0], outs[1], rtol=0, atol=0)
    ref = orp.channel_normalise(OracleVAEEncoder(sd, SMALL_O, act_dtype=torch.bfloat16).embed(imgs.cpu(), None))
    assert outs[0].shape == (5, 4, 8, 8) and rel_l2(outs[0], ref) <= 2.5e-2
    # the reference's stochastic embed_fn, seeded: same generator -> same cache
    e1 = v.embed_fn(torch.Generator(device="cuda").manual_seed(3))(imgs)
    e2 = v.embed_fn(torch.Generator(device="cuda").manual_seed(3))(imgs)
    torch.testing.assert_close(e1, e2, rtol=0, atol=0)


def test_sd3_style_vae_without_quant_convs_and_with_shift():
    """SD-v3's VAE: 16 latent channels, no (post_)quant_conv, decode(latents / scaling_factor + shift_factor)
    (models/sdv3/safe_denoiser_pipeline.py:1196)."""
    from safe_denoiser_amd.vae import SD3_VAE_CONFIG
    cfg = dict(SD3_VAE_CONFIG, block_out_channels=(64, 128), layers_per_block=1, sample_size=16)
    v = AutoencoderKL(**cfg)
    sd = {k: t for k, t in v.synthetic_state_dict(9).items() if not k.startswith("post_quant_conv")}     # the checkpoint has none
    v.load_state_dict(sd)
    lat = torch.randn(2, 16, 8, 8, generator=torch.Generator().manual_seed(4)) * 1.5
    got = torch.from_numpy(v.decode_latents(lat.cuda()))
    osd = dict(sd); osd["post_quant_conv.weight"] = torch.eye(16).reshape(16, 16, 1, 1); osd["post_quant_conv.bias"] = torch.zeros(16)
    o = OracleVAEDecoder(osd, dict(latent_channels=16, block_out_channels=(64, 128), layers_per_block=1), act_dtype=torch.bfloat16)
    ref = (o.decode(lat / 1.5305 + 0.0609) / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)
    assert got.shape == (2, 16, 16, 3) and float((got - ref).abs().mean()) <= 5e-3


def test_decoder_with_more_than_4096_attention_tokens():
    """1024 x 1024 images give the mid-block attention 16384 tokens: the score matrix is processed in blocks of 4096
    queries (sdn_softmax_rows with n = 16384).  Small channel counts keep the CPU oracle affordable."""
    cfg = dict(block_out_channels=(64, 64), layers_per_block=1, sample_size=256)         # latent side 128 -> 16384 tokens
    v = AutoencoderKL(**cfg)
    sd = v.synthetic_state_dict(5)
    v.load_state_dict(sd)
    z = torch.randn(1, 4, 128, 128, generator=torch.Generator().manual_seed(2))
    img = v.decode(z.cuda()).sample
    ref = OracleVAEDecoder(sd, dict(block_out_channels=(64, 64), layers_per_block=1), act_dtype=torch.bfloat16).decode(z)
    r = rel_l2(img, ref)
    print(f"VAE decoder with 16384 attention tokens: rel L2 vs bf16-emulating oracle {r:.3e}")
    assert img.shape == (1, 3, 256, 256) and r <= 2.5e-2


def test_any_batch_through_the_entry_points():
    """sdn_vae_decode / sdn_vae_encode cut batches above their per-invocation bound (8 images; 32-bit DMA offsets) into chunks
    themselves: a batch of 19 equals the concatenation of its images decoded / encoded in smaller calls, bit for bit."""
    v = AutoencoderKL(**SMALL)
    v.load_state_dict(v.synthetic_state_dict(9, with_encoder=True))
    g = torch.Generator().manual_seed(1)
    z = torch.randn(19, 4, v.latent_size, v.latent_size, generator=g).cuda()
    full = v.decode(z).sample
    parts = torch.cat([v.decode(z[lo:lo + 5]).sample for lo in range(0, 19, 5)])
    torch.testing.assert_close(full, parts, rtol=0, atol=0)
    x = torch.randn(11, 3, full.shape[-1], full.shape[-1], generator=g).cuda()
    m_full = v.encode(x).latent_dist.parameters
    m_parts = torch.cat([v.encode(x[lo:lo + 3]).latent_dist.parameters for lo in range(0, 11, 3)])
    torch.testing.assert_close(m_full, m_parts, rtol=0, atol=0)
