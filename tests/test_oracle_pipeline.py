"""KATs for the loop oracle on a tiny UNet (CPU): the per-prompt draw order of row S2 and the gating variants."""
import torch

from oracle import pipeline as opipe
from oracle import repellency as orp
from oracle import schedulers as osch
from oracle.unet import OracleUNet
from safe_denoiser_amd.unet import UNet2DConditionModel

TINY = dict(block_out_channels=(64, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), layers_per_block=1,
            attention_head_dim=1, cross_attention_dim=64, sample_size=8)
TINY_O = dict(block_out_channels=(64, 64), level_has_attn=(True, False), layers_per_block=1, n_heads=1, cross_dim=64,
              sample_size=8)


class Tape:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.count = {}

    def __call__(self, p, shape):
        self.count[p] = self.count.get(p, 0) + 1
        return torch.randn(shape, generator=self.g)


def _setup():
    u = UNet2DConditionModel(text_len=5, **TINY)
    sd = u.synthetic_state_dict(3)
    unet = OracleUNet(sd, TINY_O)
    g = torch.Generator().manual_seed(1)
    text = torch.randn(2, 5, 64, generator=g)
    refs = orp.channel_normalise(torch.randn(6, 4, 8, 8, generator=g))
    return unet, text, refs


def test_draw_order_counts():
    unet, text, refs = _setup()
    steps = 10                                         # t = 901, 801, 701, ... ; window 780 <= t: two steps
    base = dict(flavour="threshold", proj_refs=refs, sigma=3.15, scale=0.33, beta_threshold_margin=0.0)
    # never negate: 1 (latents) + 10 (variance) + 2 (discarded probe draws)
    tape = Tape(0)
    _, st = opipe.denoise_one(unet, osch.DDPM(), text, 0, tape, num_inference_steps=steps,
                              repel=dict(base, beta_threshold=1e9))
    assert st["renoise_draws"] == 0 and tape.count[0] == 1 + steps + 2
    # always negate: + 2 re-noise draws
    tape = Tape(0)
    _, st = opipe.denoise_one(unet, osch.DDPM(), text, 0, tape, num_inference_steps=steps,
                              repel=dict(base, beta_threshold=-1e9))
    assert st["renoise_draws"] == 2 and tape.count[0] == 1 + steps + 2 + 2
    # no processor: 1 + 10 ; DDIM: only the latents (+ re-noise draws when a processor fires)
    tape = Tape(0)
    opipe.denoise_one(unet, osch.DDPM(), text, 0, tape, num_inference_steps=steps)
    assert tape.count[0] == 1 + steps
    tape = Tape(0)
    _, st = opipe.denoise_one(unet, osch.DDIM(), text, 0, tape, num_inference_steps=steps,
                              repel=dict(base, beta_threshold=-1e9))
    assert tape.count[0] == 1 + 2 and st["renoise_draws"] == 2


def test_time_variant_renoises_the_negative_score():
    """`_time` pipelines call conditioning(beta_threshold=False): with the threshold module the returned "x_0_hat" is
    the negative score (SURVEY.md 3.2 interaction trap) and it is re-noised unconditionally in 800 <= t <= 1000."""
    unet, text, refs = _setup()
    tape = Tape(0)
    lat, st = opipe.denoise_one(unet, osch.DDPM(), text, 0, tape, num_inference_steps=10, variant="time",
                                repel=dict(flavour="threshold", proj_refs=refs, sigma=3.15, scale=0.33))
    assert st["renoise_draws"] == 2 and torch.isfinite(lat).all()
    tape2 = Tape(0)
    lat2, _ = opipe.denoise_one(unet, osch.DDPM(), text, 0, tape2, num_inference_steps=10, variant="time",
                                repel=dict(flavour="fast", proj_refs=refs, scale=0.33))
    assert not torch.allclose(lat, lat2)               # the fast module re-noises the repelled x0 instead
