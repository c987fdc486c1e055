"""The host rule that decides when a UNet forward runs as an aligned part + a concurrent tail (unet._tail_split_of; the GPU tests
`test_tail_split_forward_is_the_single_forward` / `test_a_batch_a_few_prompts_over_whole_waves_...` hold the bits).  No GPU here."""
from safe_denoiser_amd.mmdit import SD3Transformer2DModel
from safe_denoiser_amd.unet import UNet2DConditionModel


def _net(rep, on=True, cls=UNet2DConditionModel):
    u = object.__new__(cls)                      # the rule reads three attributes; no engine handle needed
    u.latent_repeat, u.tail_split = rep, on
    return u


def test_rule_splits_only_a_small_tail_over_a_multiple_of_64_samples():
    u = _net(3)
    # 3 branches: 64 prompts = 192 samples is the quantum; the 8-rank 515-prompt job gives three ranks 65 prompts
    assert u._tail_split_of(3 * 65) == (64, 1)
    assert u._tail_split_of(3 * 67) == (64, 3)                      # one GPU: the last batch of 515 with its 3 leftover prompts folded in
    assert u._tail_split_of(3 * 68) == (64, 4)
    assert u._tail_split_of(3 * 69) is None                         # 15 samples: above TAIL_MAX_SAMPLES the batch stays whole
    assert u._tail_split_of(3 * 130) == (128, 2)
    for p in (1, 3, 63, 64, 128, 192):
        assert u._tail_split_of(3 * p) is None, p
    u2 = _net(2)                                                    # 2 branches: 32 prompts = 64 samples
    assert u2._tail_split_of(2 * 33) == (32, 1) and u2._tail_split_of(2 * 97) == (96, 1) and u2._tail_split_of(2 * 32) is None
    assert u2._tail_split_of(2 * 39) is None


def test_rule_is_off_by_default_without_shared_latents_and_for_the_other_plans():
    assert _net(3, on=False)._tail_split_of(195) is None
    assert _net(1)._tail_split_of(65) is None                       # branch-major rows cannot be cut per prompt without the repeat count
    assert _net(3, cls=SD3Transformer2DModel)._tail_split_of(195) is None
    for pm_r in (_net(3)._tail_split_of(3 * p) for p in range(1, 400)):
        if pm_r is not None:
            pm, r = pm_r
            assert pm % 64 == 0 and 1 <= r and 3 * r <= UNet2DConditionModel.TAIL_MAX_SAMPLES


def test_default_is_off_on_the_model_and_on_in_the_pipeline():
    from safe_denoiser_amd.pipeline import SafeDenoiserPipeline
    from safe_denoiser_amd.schedulers import DDPMScheduler
    u = UNet2DConditionModel(text_len=77, latent_repeat=3)          # host-side create only (libsdn loads without a GPU)
    assert u.tail_split is False and u._tail_split_of(195) is None
    assert u.set_tail_split(True)._tail_split_of(195) == (64, 1)
    assert SafeDenoiserPipeline(u, DDPMScheduler()).tail_split is True


def test_batches_above_the_plan_limit_are_cut_into_whole_wave_chunks():
    """unet.max_samples / _chunks_of: one launch plan addresses its operands with 31-bit offsets -> 273 samples for SD-v1.4; above that
    the forward used to fail (SDN_E_INVALID), now it runs as chunks of 64 x k samples, a small remainder beside them."""
    u = UNet2DConditionModel(text_len=77, latent_repeat=3)
    assert u.max_samples() == 273
    assert u._chunks_of(3 * 64) is None and u._chunks_of(3 * 91) is None                      # fits: one plan (no tail rule: split off)
    assert u._chunks_of(3 * 92) == [(0, 64, False), (64, 28, False)]
    assert u._chunks_of(3 * 128) == [(0, 64, False), (64, 64, False)]
    assert u._chunks_of(3 * 130) == [(0, 64, False), (64, 64, False), (128, 2, False)]
    u.set_tail_split(True)
    assert u._chunks_of(3 * 65) == [(0, 64, False), (64, 1, True)]                            # the tail rule, in the same form
    assert u._chunks_of(3 * 130) == [(0, 64, False), (64, 64, False), (128, 2, True)]
    assert u._chunks_of(3 * 133) == [(0, 64, False), (64, 64, False), (128, 5, False)]        # 15 samples: a chunk of its own, in line
    for p in range(1, 700):
        ch = u._chunks_of(3 * p)
        if ch is not None:
            assert [c[0] for c in ch] == [sum(c[1] for c in ch[:i]) for i in range(len(ch))] and sum(c[1] for c in ch) == p
            assert all(3 * c[1] <= 273 for c in ch) and all(c[1] % 64 == 0 for c in ch[:-1])
    plain = UNet2DConditionModel(text_len=77)                                                  # latent_repeat 1: rows are cut directly
    assert plain._chunks_of(273) is None and plain._chunks_of(384) == [(0, 256, False), (256, 128, False)]
    two = UNet2DConditionModel(text_len=77, latent_repeat=2)
    assert two._chunks_of(2 * 136) is None and two._chunks_of(2 * 137) == [(0, 128, False), (128, 9, False)]
    assert _net(3, cls=SD3Transformer2DModel)._chunks_of(3000) is None                         # other plans keep their own rules
    x3 = UNet2DConditionModel(text_len=77, latent_repeat=3, precision="bf16x3")               # fp32 storage: 3/4 of the 16-bit limit
    assert x3.max_samples() == 204 and x3._chunks_of(3 * 68) is None and x3._chunks_of(3 * 70) == [(0, 64, False), (64, 6, False)]
    assert SD3Transformer2DModel(sample_size=64).max_samples() == 170 and SD3Transformer2DModel(sample_size=128).max_samples() == 42
