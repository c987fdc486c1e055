"""Host logic of the per-step precision schedule (SafeDenoiserPipeline.hi_steps): which steps run on the precise plan."""
import pytest

from safe_denoiser_amd._lib import SdnError
from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
from safe_denoiser_amd.schedulers import DDPMScheduler


class _U:                      # the two attributes the constructor compares
    text_len, latent_repeat = 77, 3

    class config:
        sample_size, in_channels, cross_attention_dim = 64, 4, 768


def _pipe(spec):
    return SafeDenoiserPipeline(_U(), DDPMScheduler(), unet_hi=_U(), precision_schedule=spec)


def _ts():
    s = DDPMScheduler()
    s.set_timesteps(50)
    return [int(t) for t in s.timesteps]


def test_schedule_forms():
    ts = _ts()
    assert ts[0] == 981 and ts[-1] == 1 and len(ts) == 50
    win = [i for i, t in enumerate(ts) if 780 <= t <= 1000]
    assert win == list(range(11))                                   # the 11 window steps of the README configuration
    on = lambda spec: [i for i, h in enumerate(_pipe(spec).hi_steps(ts, "t", 780, 1000)) if h]
    assert on("all") == list(range(50)) and on("none") == []
    assert on({"first": 3}) == [0, 1, 2]
    assert on({"last": 2}) == [48, 49]
    assert on({"window": True, "last": 1}) == win + [49]
    assert on({"steps": [7, 30], "first": 1}) == [0, 7, 30]
    assert on([i % 2 == 0 for i in range(50)]) == list(range(0, 50, 2))
    assert on(lambda i, t, w: w and t > 900) == [i for i, t in enumerate(ts) if t > 900]
    assert on({"first": 80}) == list(range(50))                     # clamps
    # step-index windows (the *_threshold.py / sd_threshold_time variants)
    assert [i for i, h in enumerate(_pipe({"window": True}).hi_steps(ts, "i", 0, 4)) if h] == [0, 1, 2, 3, 4]


def test_schedule_errors():
    ts = _ts()
    with pytest.raises(SdnError):
        _pipe([True] * 49).hi_steps(ts)
    with pytest.raises(SdnError):
        _pipe({"frist": 3}).hi_steps(ts)
    with pytest.raises(SdnError):
        SafeDenoiserPipeline(_U(), DDPMScheduler(), precision_schedule="all")           # no precise plan given
    with pytest.raises(SdnError):
        SafeDenoiserPipeline(_U(), DDPMScheduler(), unet_hi=_U())                        # ... and the converse

    class Other(_U):
        latent_repeat = 2
    with pytest.raises(SdnError):
        SafeDenoiserPipeline(_U(), DDPMScheduler(), unet_hi=Other(), precision_schedule="all")


def test_pipeline_keeps_one_sibling_plan_per_net_and_branch_count():
    """`_sibling` (the handles the per-branch precision of the precise steps runs on: the precise plan with two branches, the 16-bit plan
    with one): same architecture / storage type / precision as the net, another latent_repeat, the SAME weights object, created once."""
    import torch
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import DDPMScheduler
    from safe_denoiser_amd.unet import UNet2DConditionModel
    small = dict(block_out_channels=(320, 640), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), layers_per_block=1,
                 attention_head_dim=8, cross_attention_dim=768, sample_size=16)
    lo = UNet2DConditionModel(text_len=77, dtype=torch.float16, latent_repeat=3, **small)
    hi = UNet2DConditionModel(text_len=77, precision="bf16x3", latent_repeat=3, **small)
    lo._weights, hi._weights = object(), object()                    # (stand-ins: no GPU here)
    pipe = SafeDenoiserPipeline(lo, DDPMScheduler(), unet_hi=hi, precision_schedule={"window": True})
    assert pipe.dead_branch_lo is True
    h2, l1 = pipe._sibling(hi, 2), pipe._sibling(lo, 1)
    assert (h2.latent_repeat, h2.precision, h2.dtype, h2._weights is hi._weights) == (2, "bf16x3", torch.float32, True)
    assert (l1.latent_repeat, l1.precision, l1.dtype, l1._weights is lo._weights) == (1, None, torch.float16, True)
    assert vars(h2.config) == vars(hi.config) and h2.weight_bytes == hi.weight_bytes and l1.weight_bytes == lo.weight_bytes
    assert pipe._sibling(hi, 2) is h2 and pipe._sibling(lo, 1) is l1
    new_w = object(); lo._weights = new_w                            # reloaded weights are picked up at the next use
    assert pipe._sibling(lo, 1)._weights is new_w
