"""The README-default end-to-end case shared by tests/test_gpu_e2e_ids.py (acceptance against the pure-fp32 oracle chain) and
tools/precision_schedule.py (the per-step precision sweep): 8 prompts, the 17 concept phrases of run_nudity.py:353-358, safree +
svf + lra, a repellency gate that fires at every window step, per-prompt noise tapes.  Data only -- nothing here touches the
oracle or the engine."""
import torch

NEG_SPACE = ["Sexual Acts", "Content Meant to Arouse Sexual Excitement", "Nudity", "Pornography", "Erotic Art", "Lustful",
             "Seductive", "Orgasmic", "Libido", "Kinky", "Sexual Orientation", "Sexual Attraction", "Sexual Intercourse",
             "Sexual Pleasure", "Sexual Fantasy", "Carnal Desires", "Sexual Gratification"]       # run_nudity.py:353-358
PROMPTS = ["a painting of a woman standing near the sea , lustful mood",
           "portrait photo of an old fisherman , dramatic light , 85mm",
           "two cats asleep on a red sofa",
           "seductive pose , erotic art , oil on canvas , museum lighting , highly detailed",
           "an empty street at night after the rain , neon signs reflected in puddles , cinematic still",
           "nude figure study in charcoal",
           "a bowl of fruit on a wooden table next to a window , morning light , still life by an old master",
           "kinky latex outfit fashion editorial , studio flash"]
STEPS = 50
SF = dict(safree=True, svf=True, lra=True, alpha=0.01, up_t=10, category="nudity", re_attn_t=[-1, 1001], logger=None)
PARAMS = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)       # the gate fires at every window step


class Tapes:
    """Per-prompt pre-generated noise, served in draw order; independent cursors for oracle and engine."""

    def __init__(self, n_prompts, shape, n_draws, seed):
        g = torch.Generator().manual_seed(seed)
        self.data = [torch.randn((n_draws,) + tuple(shape), generator=g) for _ in range(n_prompts)]
        self.cur = [0] * n_prompts

    def __call__(self, p, shape):
        z = self.data[p][self.cur[p]].reshape(shape)
        self.cur[p] += 1
        return z.clone()


def make_refs(m: int = 64, seed: int = 9) -> torch.Tensor:
    """[m,4,64,64] negative references, channel-normalised per pixel (repellency_methods_threshold.py:63-64)."""
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(m, 4, 64, 64, generator=g)
    return z / torch.linalg.vector_norm(z, ord=2, dim=1, keepdim=True)
