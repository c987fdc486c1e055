"""The denoising loop with per-step repellency projection -- batched over prompts.

Call surface: the reference's `ModifiedSafreeDiffusionPipeline_Rep.__call__`
(models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:352-375,598) -- same argument order, names and
defaults, so `run_nudity.py:439-460` runs verbatim: `pipe(prompt, num_images_per_prompt=1, guidance_scale=, num_inference_steps=,
negative_prompt=, negative_prompt_space=, height=, width=, generator=, repellency_processor=, safree_dict=, **SLD config)`
-> list of PIL images (`return_latents=True` = the parity tap, :585-586).  `from_pretrained(local_dir)` builds the stack from a
diffusers-layout checkpoint directory (run_nudity.py:104-131).  With prompt strings the engine's CLIPTextModel encodes them and
the SAFREE block (:458-486) runs inside the call; alternatively pass CLIP hidden states as `prompt_embeddings`
([2P,77,768] = chunk(2) -> [P uncond | P text]; [3P,...] = + safety concept for the SLD families).  With a `vae`
(safe_denoiser_amd.vae.AutoencoderKL) the call ends like the reference's (:588-596): `decode_latents` -> PIL images
(`output_type="pil"`), NHWC float numpy in [0,1] ("np"), or the uint8 NHWC tensor numpy_to_pil would build, left on the device
("uint8").

What differs from the reference by design (results per sample are the same):
  * P prompts are denoised together (the reference is hard-wired to batch 1); every prompt keeps its OWN
    torch.Generator so its random stream -- latents, the discarded variance draw of the x0 probe, the conditional
    re-noise draw, the step's variance draw, in that order (row S2) -- is exactly the reference's;
  * all per-element math is in libsdn kernels; the only host<->device traffic in the loop is ONE readback of the
    P is_negation flags per repellency-window step (the reference syncs three times per step per prompt), needed
    only because the number of randn draws -- hence each torch.Generator's offset -- depends on the flag;
  * the per-prompt random draws of a step are one launch per kind (rng.py: bit-identical to torch.randn per generator)
    from generator states uploaded once per call and advanced on the device.
"""
from __future__ import annotations

import itertools
from typing import Callable, Optional, Sequence

import torch

from . import _lib
from .schedulers import DDIMScheduler, DDPMScheduler

# gating variants of the reference's pipeline files: (window kind, lo, hi,
# beta_threshold kwarg, honours is_negation)
VARIANTS = {
    "threshold_time": ("t", 780, 1000, True, True),     # {safree,sld}_*_threshold_time.py  (north-star path)
    "time": ("t", 800, 1000, False, False),             # *_time.py (the SLD file's default lower bound is 780, :380-382)
    "threshold": ("i", 0, 50, True, True),              # *_threshold.py: window HARD-CODED, kwargs ignored (:430-431)
    "sd_threshold_time": ("i", 0, 11, True, True),      # modified_stable_diffusion_pipeline_threshold_time.py: step INDEX
                                                        # window, negation_warmup_start = lower / _end = upper bound (:430-431,551)
    "plain": (None, 0, 0, False, True),                 # modified_{safree,sld}_diffusion_pipeline.py: every step
}


# text-contents versions handed to sdn_unet_set_text_version: ONE counter for the process, so that two pipelines over one UNet (or a
# pipeline built after another was freed, whose text buffers the caching allocator may hand back at the same addresses) can never
# declare the same number for different text
_TEXT_VERSIONS = itertools.count(1)


SLD_SAFETY_CONCEPT = ("hate, harassment, violence, suffering, humiliation, harm, suicide, sexual, nudity, bodily fluids, blood, "
                      "obscene gestures, illegal activity, drug use, theft, vandalism, weapons, child abuse, brutality, cruelty")


class SafeDenoiserPipeline:
    GRAPH_MAX_BATCH = 16      # UNet rows (branches x prompts) up to which a forward is launch-bound and replayed as a hipGraph

    def __init__(self, unet, scheduler, variant: str = "threshold_time", vae=None, use_graphs: Optional[bool] = None,
                 text_encoder=None, tokenizer=None, split_k: bool = False, elide_dead_branch: bool = False,
                 unet_hi=None, precision_schedule=None):
        if variant not in VARIANTS:
            raise KeyError(f"unknown variant {variant}; have {sorted(VARIANTS)}")
        self.unet, self.scheduler, self.variant, self.vae = unet, scheduler, variant, vae
        # Precision schedule (round 5): `unet_hi` is a second plan over the SAME weights in a precision mode (bf16x3 / fp32) and
        # `precision_schedule` says which steps of the loop run on it; every other step runs on `unet` (the 16-bit plan).  The
        # reference runs every step in fp32 (run_nudity.py:277); the schedule spends the precise plan where the final latents
        # are sensitive to the step's rounding (tools/precision_schedule.py measures that) -- see `hi_steps()` for the forms.
        self.unet_hi, self.precision_schedule = unet_hi, precision_schedule
        if (unet_hi is None) != (precision_schedule is None):
            raise _lib.SdnError("unet_hi and precision_schedule go together")
        if unet_hi is not None:
            a, b = unet.config, unet_hi.config
            if (a.sample_size, a.in_channels, a.cross_attention_dim) != (b.sample_size, b.in_channels, b.cross_attention_dim) \
                    or getattr(unet, "latent_repeat", 1) != getattr(unet_hi, "latent_repeat", 1) or unet.text_len != unet_hi.text_len:
                raise _lib.SdnError("unet_hi must be the same architecture / latent_repeat / text_len as unet")
        self.use_graphs = use_graphs          # None = automatic: on for small batches (the reference's one-prompt calls)
        # optional front end: safe_denoiser_amd.clip.CLIPTextModel + the caller's CLIPTokenizer
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        # split_k=True: small batches additionally run their long, thin GEMMs in split-K form (lower single-prompt latency;
        # a prompt's result then depends, in the last fp32 bits, on how many prompts share its batch)
        self.split_k = split_k
        # `lra` makes the reference run a THIRD guidance branch ([uncond | text' | text]) whose output it then discards
        # (`noise_pred_uncond, noise_pred_text, _ = noise_pred.chunk(3)`, ...threshold_time.py:542-544); samples do not interact
        # inside the UNet, so the two live branches give the same bits without it.  Off by default: the engine then does the
        # reference's work, branch for branch.  True = skip the dead branch (same images, 2/3 of the UNet work).
        self.elide_dead_branch = elide_dead_branch
        self.vae_scale_factor = 8
        self.last_stats = {}
        self.last_safree = None
        self._bufs = {}
        self._rngs = {}
        self.batched_rng = True    # False: per-prompt torch.randn calls (the draws are the same bits either way)
        # a batch a few prompts over whole waves of tiles (65 prompts x 3 branches on three of eight ranks of the 515-prompt job) runs
        # its aligned part and its tail as two concurrent forwards (unet.UNet2DConditionModel._tail_split_of): same bits on the
        # 16-bit plans, -3.6 % per step at 65 prompts.  False: always one forward per step.
        self.tail_split = True
        # Precision per (step, BRANCH) (round 5): with `lra` the third guidance branch is computed and discarded (...threshold_time.py:542-544),
        # so its precision cannot reach the latents.  At the steps the schedule sends to `unet_hi`, the two LIVE branches run on the
        # precise plan and the discarded one on the 16-bit plan -- the reference's work, branch for branch, with the precise arithmetic
        # only where a result is read.  False: all three branches on `unet_hi` at those steps.
        self.dead_branch_lo = True
        self._sib = {}
        self.device_flags = True   # False: read the is_negation flags back at every window step (rounds 1-4; same bits either way)
        self.batched_safree = True  # False: the SAFREE projection prompt by prompt (safree.prepare), as the reference runs it
        self.record_den = False    # diagnostics: keep each window step's denominators (device tensors, no sync) in last_stats
        # SLDPipeline._safety_text_concept: the default of the third-party base class the reference's SLD pipelines inherit
        # (python_sld == 1.0.10, requirements.txt:16; not in /root/reference -- restated from the published package)
        self.safety_concept = SLD_SAFETY_CONCEPT

    # ------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_dir: str, scheduler=None, torch_dtype=torch.bfloat16, variant: str = "threshold_time",
                        latent_repeat: int = 1, tokenizer=None, weights_variant: Optional[str] = None, precision: Optional[str] = None,
                        device="cuda", text_encoder_precision: Optional[str] = "bf16x3", **kwargs):
        """`pipeline_func.from_pretrained(model_id, scheduler=scheduler, torch_dtype=weight_dtype, revision="fp16")`
        (run_nudity.py:104-122) for a LOCAL diffusers-layout directory: unet / vae / text_encoder weights (safetensors or
        .bin) are packed into the engine's layouts, `scheduler/scheduler_config.json` is honoured when no scheduler object is
        passed (DDPMScheduler.from_pretrained, :108 -- including its clip_sample) and the tokenizer is
        loaded when its vocabulary files are present.  `torch_dtype`: bf16 / fp16 storage, or float32 = the fp32 plan
        (`precision="bf16x3"` = fp32 storage with split-operand contractions); the VAE and text encoder take the 16-bit type.
        `precision="scheduled"` (round 5) = the fp32 RESULT at 1.4 x the 16-bit cost: the fp16 plan and the bf16x3 plan over the
        same weights, the bf16x3 one on the steps inside the repellency window (`precision_schedule=`, default {"window": True};
        DESIGN 10.1), text encoder bf16x3.
        `variant` = the gating variant (SD_FUNCTIONS[erase_id], driver.ERASE_IDS); `latent_repeat` = guidance branches that
        share their latents (2 = CFG, 3 = lra / SLD).  `revision` and other hub arguments are accepted and ignored."""
        import os

        from . import checkpoint as ck
        from .clip import CLIPTextModel
        from .unet import UNet2DConditionModel
        from .vae import AutoencoderKL
        if not os.path.isdir(model_dir):
            raise FileNotFoundError(f"{model_dir}: not a local directory (there is no hub access; pass a diffusers-layout checkpoint directory)")
        if scheduler is None:
            scheduler = DDPMScheduler.from_pretrained(model_dir, subfolder="scheduler")
        dt16 = torch_dtype if torch_dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
        ucfg = ck.unet_kwargs(ck.read_config(os.path.join(model_dir, "unet")))
        usd = ck.load_weights(os.path.join(model_dir, "unet"), weights_variant)
        unet_hi = schedule = None
        if precision == "scheduled":
            schedule = kwargs.pop("precision_schedule", None) or {"window": True}
            unet = UNet2DConditionModel(dtype=torch.float16, latent_repeat=latent_repeat, **ucfg)     # fp16: a bf16 step costs 8 x its error
            unet_hi = UNet2DConditionModel(latent_repeat=latent_repeat, precision="bf16x3", **ucfg)
            unet_hi.load_state_dict(usd, device=device)
        else:
            unet = UNet2DConditionModel(dtype=torch_dtype, latent_repeat=latent_repeat, precision=precision, **ucfg)
        unet.load_state_dict(usd, device=device)
        del usd
        vae = enc = None
        if os.path.isdir(os.path.join(model_dir, "vae")):
            vae = AutoencoderKL(dtype=dt16, **ck.vae_kwargs(ck.read_config(os.path.join(model_dir, "vae"))))
            vae.load_state_dict(ck.load_weights(os.path.join(model_dir, "vae"), weights_variant), device=device)
        if os.path.isdir(os.path.join(model_dir, "text_encoder")):
            # the reference loads the text encoder in the pipeline's dtype (run_nudity.py:277: fp32): an fp32 / bf16x3 UNet gets the
            # text encoder in the same precision mode -- its states feed every cross-attention and the SAFREE decisions
            # ... and, round 5, a 16-bit UNet too gets the text encoder in bf16x3 by default (`text_encoder_precision`; None = the
            # UNet's storage type): a bf16 text encoder flips a SAFREE trigger-token decision for 1 prompt in 8 against the fp32 chain
            # (tests/test_gpu_e2e_ids.py), a categorical divergence, and the encoder is < 1 % of a call's time
            if unet_hi is not None:
                enc_kw = dict(precision="bf16x3")
            elif unet.precision is not None:
                enc_kw = dict(precision=unet.precision)
            else:
                enc_kw = dict(precision=text_encoder_precision) if text_encoder_precision else dict(dtype=dt16)
            enc = CLIPTextModel(**enc_kw, **ck.clip_kwargs(ck.read_config(os.path.join(model_dir, "text_encoder"))))
            enc.load_state_dict(ck.load_weights(os.path.join(model_dir, "text_encoder"), weights_variant), device=device)
        if tokenizer is None:
            tokenizer = ck.load_tokenizer(model_dir)
        return cls(unet, scheduler, variant=variant, vae=vae, text_encoder=enc, tokenizer=tokenizer, unet_hi=unet_hi,
                   precision_schedule=schedule)

    def to(self, *args, **kwargs):
        """`pipe.to(device)` of the reference's load_sd (run_nudity.py:131): the engine's weights already live on the GPU."""
        return self

    def _rng_for(self, dev, numel: int):
        if not self.batched_rng:
            return None
        from .rng import BatchedNormal
        key = (str(dev), int(numel))
        r = self._rngs.get(key)
        if r is None:
            r = self._rngs[key] = BatchedNormal(dev, numel)
        return r

    def _noise(self, noise_fn, generators, p: int, shape, device):
        if noise_fn is not None:
            return noise_fn(p, shape).to(device=device, dtype=torch.float32)
        return torch.randn(shape, generator=generators[p], device=device, dtype=torch.float32)

    # keyword arguments the reference's __call__ variants read out of **kwargs (window bounds, the SLD SafetyConfig splat)
    _KNOWN_KWARGS = ("negation_warmup_start", "negation_warmup_end", "negation_warmup_steps", "sld_guidance_scale",
                     "sld_warmup_steps", "sld_threshold", "sld_momentum_scale", "sld_mom_beta", "safree")

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None,
                 negative_prompt_space=None, num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents=None, output_type: Optional[str] = "pil", return_dict: bool = True, callback=None,
                 callback_steps: Optional[int] = 1, prompt_ids=None,
                 prompt_embeddings: Optional[torch.Tensor] = None, return_latents: bool = False,
                 repellency_processor=None, safree_dict=None,
                 rescaled_text_embeddings: Optional[torch.Tensor] = None, beta_adjusted=None,
                 noise_fn: Optional[Callable] = None, **kwargs):
        """Argument order, names and defaults of the reference's __call__ (...threshold_time.py:352-375), so its call site
        (run_nudity.py:439-460) works verbatim: `return_latents` defaults to False (images come back, :585-596).
        `return_dict` is accepted and, as in the reference (:598: `return image`), does not change what is returned.
        Honoured: `callback(i, t, latents)` every `callback_steps` (:580-582).  Rejected loudly instead of silently dropped:
        `num_images_per_prompt` != 1 (the reference's repellency block is batch-1: `denominator.item()`, repellency_methods_
        threshold.py:348 -- batch prompts instead), `eta` != 0 with DDIM (DDPM never receives eta: prepare_extra_step_kwargs),
        `prompt_ids`, and keyword arguments no variant of the reference reads."""
        _lib.require_gpu()
        if num_images_per_prompt not in (None, 1):
            raise NotImplementedError("num_images_per_prompt > 1: the reference's repellency block handles one latent per call "
                                      "(denominator.item(), repellency_methods_threshold.py:348); pass the prompt "
                                      "num_images_per_prompt times with one generator each instead")
        if eta and isinstance(self.scheduler, DDIMScheduler):
            raise NotImplementedError("DDIM with eta != 0 is not implemented (the reference's live scheduler is DDPM, which "
                                      "never receives eta)")
        if prompt_ids is not None:
            raise NotImplementedError("prompt_ids (the SLD pipelines' embedding-space prompt) is not on the benchmarked path")
        if callback is not None and (not isinstance(callback_steps, int) or callback_steps <= 0):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        unknown = sorted(k for k in kwargs if k not in self._KNOWN_KWARGS)
        if unknown:
            raise TypeError(f"SafeDenoiserPipeline.__call__: unknown keyword arguments {unknown}")
        if output_type is None:
            output_type = "np"
        sf = dict(safree=False, svf=False, lra=False, re_attn_t=(-1, -1), alpha=0.0, up_t=10, category="nudity", logger=None)
        if safree_dict:
            sf.update(safree_dict)
        # SLD family (modified_sld_pipeline*.py): third branch = safety concept, guidance eq. 3-8 with momentum state
        # (enable_safety_guidance = sld_guidance_scale >= 1, modified_sld_pipeline_threshold_time.py:402-405)
        sld = None
        if kwargs.get("sld_guidance_scale", 0) and kwargs["sld_guidance_scale"] >= 1:
            sld = dict(scale=float(kwargs["sld_guidance_scale"]), warmup=int(kwargs.get("sld_warmup_steps", 10)),
                       thr=float(kwargs.get("sld_threshold", 0.01)), ms=float(kwargs.get("sld_momentum_scale", 0.3)),
                       mb=float(kwargs.get("sld_mom_beta", 0.4)))
        if sld:                                    # the SLD pipelines take safree_dict / negative_prompt_space into **kwargs and never
            sf.update(safree=False, svf=False, lra=False)       # read them (modified_sld_pipeline_threshold_time.py:285-311)
        n_prompts = None if prompt is None else (1 if isinstance(prompt, str) else len(prompt))
        if prompt_embeddings is None:
            if prompt is None or self.text_encoder is None or self.tokenizer is None:
                raise NotImplementedError("pass `prompt_embeddings` ([2P,77,768]), or construct the pipeline with text_encoder= "
                                          "(safe_denoiser_amd.clip.CLIPTextModel) and tokenizer= and pass `prompt` strings")
            # steps 3 of the reference's __call__ (...threshold_time.py:453-486): encode, then the SAFREE text projection
            prompt_embeddings, _ids, attn_mask = self._new_encode_prompt(prompt, negative_prompt,
                                                                         safety_concept=self.safety_concept if sld else None)
            if sf["safree"] and rescaled_text_embeddings is None:
                if negative_prompt_space is None:
                    raise _lib.SdnError("safree_dict['safree'] needs negative_prompt_space (the concept phrases)")
                prep = self._safree_prepare(prompt, prompt_embeddings, attn_mask, negative_prompt_space, sf)
                rescaled_text_embeddings = prep["rescaled_text_embeddings"]
                if sf["svf"] and beta_adjusted is None:
                    beta_adjusted = prep["beta_adjusted"]
                self.last_safree = prep
        if not return_latents:
            if self.vae is None:
                raise NotImplementedError("no VAE decoder attached: construct the pipeline with vae=AutoencoderKL(...) or "
                                          "use return_latents=True (the reference's parity tap, ...threshold_time.py:585-586)")
            if output_type not in ("pil", "np", "uint8"):
                raise _lib.SdnError("output_type must be 'pil', 'np' or 'uint8'")
        kind, lo_default, hi_default, use_beta, use_flag = VARIANTS[self.variant]
        if self.variant == "threshold":                               # the reference assigns 0 / 50 and never reads the kwargs
            lo, hi = lo_default, hi_default
        elif kind == "i":                                             # reference: i >= start and i <= end
            lo = kwargs.get("negation_warmup_start", lo_default)
            hi = kwargs.get("negation_warmup_end", hi_default)
        else:                                                         # reference: t <= start and t >= end
            hi = kwargs.get("negation_warmup_start", hi_default)
            lo = kwargs.get("negation_warmup_end", 780 if (sld and self.variant == "time") else lo_default)
        nb = 3 if ((sf["lra"] and not self.elide_dead_branch) or sld) else 2
        # one scale for the call (the reference's signature), or one per prompt: the drivers read it row by row from the prompt
        # table (run_nudity.py:390-396) and the batched engine keeps such rows in one batch (sdn_cfg_combine_rows)
        g_rows = None
        if isinstance(guidance_scale, torch.Tensor):
            guidance_scale = guidance_scale.detach().flatten().tolist()
        if isinstance(guidance_scale, (list, tuple)):
            g_rows = [float(g_) for g_ in guidance_scale]
            if not g_rows:
                raise _lib.SdnError("guidance_scale: empty list")
            guidance_scale = g_rows[0]
        if (min(g_rows) if g_rows is not None else guidance_scale) <= 1.0:
            raise NotImplementedError("guidance_scale <= 1 (no CFG) is not on the reference's benchmarked path")
        dev = torch.device("cuda", torch.cuda.current_device())
        E = prompt_embeddings.to(dev)
        if sld:
            if E.shape[0] % 3 != 0:
                raise _lib.SdnError("SLD: prompt_embeddings must be [3P,77,768]: P uncond | P text | P safety concept")
            P = E.shape[0] // 3
        else:
            if E.shape[0] % 2 != 0:
                raise _lib.SdnError("prompt_embeddings must be [2P,77,768]: P unconditional rows then P text rows")
            P = E.shape[0] // 2
        if g_rows is not None:
            if len(g_rows) != P:
                raise _lib.SdnError(f"guidance_scale: need one value per prompt ({P}), got {len(g_rows)}")
            # equal scales take the scalar kernel (same bits either way)
            g_rows = None if len(set(g_rows)) == 1 else torch.tensor(g_rows, dtype=torch.float32, device=dev)
        if n_prompts is not None and E.shape[0] != (3 if sld else 2) * n_prompts:
            raise _lib.SdnError(f"{E.shape[0]} text rows for {n_prompts} prompts: expected {(3 if sld else 2) * n_prompts} "
                                f"([uncond | text{' | safety concept' if sld else ''}])")
        s = self.unet.config.sample_size
        height = height or s * self.vae_scale_factor
        width = width or s * self.vae_scale_factor
        if (height // self.vae_scale_factor, width // self.vae_scale_factor) != (s, s):
            raise _lib.SdnError(f"this UNet plan is built for {s * 8}x{s * 8} images")
        C_ = self.unet.config.in_channels
        shape1 = (1, C_, s, s)
        D = C_ * s * s

        # text branches: [uncond | E'(or E) | (text_e when lra)]  (...threshold_time.py:525-540)
        E_plain = E if sld else self._branches(E, E, nb)
        E_safe = self._branches(rescaled_text_embeddings.to(dev), E, nb) if (sf["safree"] and rescaled_text_embeddings
                                                                             is not None) else None
        tb_plain = self.unet.prepare_text(E_plain)

        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        timesteps = [int(t) for t in sch.timesteps]

        if beta_adjusted is None or isinstance(beta_adjusted, (int, float)):
            beta_list = [beta_adjusted] * P
        else:
            beta_list = list(beta_adjusted)
            if len(beta_list) != P:
                raise _lib.SdnError(f"beta_adjusted: need one value per prompt ({P}), got {len(beta_list)}")
        gens = self._generators(generator, P, dev) if noise_fn is None else None
        # one launch per draw kind for the whole batch, bit-identical to the per-prompt torch.randn calls (rng.py); a noise
        # tape (tests) and exotic generators keep the per-prompt path
        rng = self._rng_for(dev, D) if (gens is not None and all(g_.device.type == "cuda" for g_ in gens)) else None

        def draw_into(dst, which=None, flags_dev=None):
            if rng is not None:
                rng.draw(gens, dst, which, shape1, flags_dev=flags_dev)
            else:
                for p in (range(P) if which is None else which):
                    dst[p:p + 1] = self._noise(noise_fn, gens, p, shape1, dev)

        def draw_discard():
            if rng is not None:
                rng.skip(gens)
            else:
                for p in range(P):
                    self._noise(noise_fn, gens, p, shape1, dev)

        if latents is None:
            lat = torch.empty((P, C_, s, s), dtype=torch.float32, device=dev)
            draw_into(lat)
            lat = lat * sch.init_noise_sigma if sch.init_noise_sigma != 1.0 else lat
        else:
            lat = latents.to(device=dev, dtype=torch.float32).clone()

        L = _lib.lib()
        st = _lib.stream_ptr()
        use_hi = self.hi_steps(timesteps, kind, lo, hi) if self.unet_hi is not None else [False] * len(timesteps)
        nets = ([self.unet] if not all(use_hi) else []) + ([self.unet_hi] if any(use_hi) else [])
        # a UNet built with latent_repeat = nb repeats the latents itself (and shares the branch-independent prefix)
        rep = getattr(self.unet, "latent_repeat", 1)
        if rep not in (1, nb):
            raise _lib.SdnError(f"unet.latent_repeat = {rep} but this call runs {nb} guidance branches")
        shared_latents = rep == nb
        for u_ in nets:
            if hasattr(u_, "set_tail_split"):
                u_.set_tail_split(bool(self.tail_split) and shared_latents)
            if hasattr(u_, "set_graph_mode"):
                small = nb * P <= self.GRAPH_MAX_BATCH if self.use_graphs is None else bool(self.use_graphs)
                u_.set_graph_mode(small)
                if hasattr(u_, "set_split_k"):
                    u_.set_split_k(bool(self.split_k) and small)
        # loop buffers are kept across calls (same shapes -> same addresses -> the UNet's graph cache keeps hitting);
        # the latents handed back to the caller are therefore a copy
        key = (P, nb, C_, s, dev, bool(shared_latents), tuple(tb_plain.shape), tuple(u_.dtype for u_ in nets))
        if self._bufs.get("key") != key:
            f32 = dict(dtype=torch.float32, device=dev)
            self._bufs = dict(key=key, x_in=None if shared_latents else torch.empty((nb * P, C_, s, s), **f32),
                              model_out=torch.empty((nb * P, C_, s, s), **f32), eps=torch.empty((P, C_, s, s), **f32),
                              x0=torch.empty((P, C_, s, s), **f32), noise=torch.empty((P, C_, s, s), **f32),
                              lat=torch.empty((P, C_, s, s), **f32), nxt=torch.empty((P, C_, s, s), **f32),
                              fired=torch.zeros(P, dtype=torch.int32, device=dev),
                              text=[{k_: torch.empty(tuple(tb_plain.shape), dtype=u_.dtype, device=dev)
                                     for k_ in ("plain", "safe", "mix")} for u_ in nets])
        bf = self._bufs
        # one set of text buffers per plan in use (each in its plan's storage type).  Every (re)write of a text buffer gets a fresh
        # version: the UNet recomputes the cross-attention K / V of the text only when the version it is handed changes (plain /
        # projected / per-prompt mix: a handful of changes over the 50 steps)
        texts = {}
        for u_, tb_ in zip(nets, bf["text"]):
            tb_["plain"].copy_(tb_plain if u_ is self.unet else u_.prepare_text(E_plain))
            if E_safe is not None:
                tb_["safe"].copy_(u_.prepare_text(E_safe))
            texts[id(u_)] = dict(buf=tb_, ver_plain=next(_TEXT_VERSIONS), ver_safe=next(_TEXT_VERSIONS), ver_mix=0, mix_key=None,
                                 has_ver=hasattr(u_, "set_text_version"))
        has_safe = E_safe is not None
        x_in, model_out, eps, x0, noise, nxt = bf["x_in"], bf["model_out"], bf["eps"], bf["x0"], bf["noise"], bf["nxt"]
        bf["lat"].copy_(lat)
        lat = bf["lat"]
        is_ddpm = isinstance(sch, DDPMScheduler)
        n_renoise = 0
        n_window = 0
        den_log = []
        self._last_den = None
        momentum = torch.zeros_like(eps) if sld else None
        # Sync-free window steps (round 5): the number of randn draws of a prompt depends on its is_negation flag, which rounds 1-4
        # read back once per window step.  With the generators' (seed, offset) pairs resident on the device (rng.py) the flag
        # vector itself selects the rows that draw and advance, so nothing has to come back: the torch.Generator objects are
        # brought up to date ONCE after the loop (BatchedNormal.sync_host), the draw count is summed on the device.
        dev_flags = bool(self.device_flags and rng is not None and rng.ok and use_flag
                         and hasattr(repellency_processor, "conditioning_device"))
        fired_acc = bf["fired"].zero_() if dev_flags else None
        # per-branch precision at the precise steps (see __init__): needs a discarded branch (lra, not SLD: its third branch is live), both
        # plans in this call, shared latents, and the engine's own UNet class on both sides
        from .unet import UNet2DConditionModel
        dead_lo = bool(self.dead_branch_lo and self.unet_hi is not None and nb == 3 and sf["lra"] and not sld and shared_latents
                       and any(use_hi) and not all(use_hi) and type(self.unet) is UNet2DConditionModel
                       and type(self.unet_hi) is UNet2DConditionModel)
        hi2 = lo1 = None
        n_dead_lo = 0
        if dead_lo:
            hi2, lo1 = self._sibling(self.unet_hi, 2), self._sibling(self.unet, 1)
            for sib in (hi2, lo1):
                sib.set_tail_split(bool(self.tail_split))
                sib.set_graph_mode(nb * P <= self.GRAPH_MAX_BATCH if self.use_graphs is None else bool(self.use_graphs))

        try:
            for i, t in enumerate(timesteps):
                u = self.unet_hi if use_hi[i] else self.unet
                tx = texts[id(u)]
                if not shared_latents:
                    x_in.view(nb, P, C_, s, s).copy_(lat)                               # cat([latents] * nb)
                # which prompts see the SAFREE-projected text at this step (...threshold_time.py:525-532); with the
                # self-validation filter the step count is PER PROMPT, so a batch can be mixed
                if not has_safe:
                    safe_p = [False] * P
                elif sf["svf"]:
                    safe_p = [ba is not None and i <= ba for ba in beta_list]
                else:
                    safe_p = [sf["re_attn_t"][0] <= i <= sf["re_attn_t"][1]] * P

                def pick_text(tx):                                                      # this step's text operand of one plan: (buffer, version)
                    tbuf = tx["buf"]
                    if all(safe_p):
                        return tbuf["safe"], tx["ver_safe"]
                    if not any(safe_p):
                        return tbuf["plain"], tx["ver_plain"]
                    if tx["mix_key"] != tuple(safe_p):                                  # the set of projected prompts changed: rebuild the mix
                        pick = torch.tensor(safe_p * nb, device=dev)[:, None, None]
                        tbuf["mix"].copy_(torch.where(pick, tbuf["safe"], tbuf["plain"]))
                        tx["mix_key"] = tuple(safe_p)
                        tx["ver_mix"] = next(_TEXT_VERSIONS)
                    return tbuf["mix"], tx["ver_mix"]

                tb, ver = pick_text(tx)
                if dead_lo and use_hi[i]:
                    # live branches [uncond | text'] on the precise plan, the discarded third on the 16-bit plan; rows are branch-major,
                    # so both operands and both outputs are contiguous row ranges of the plans' own text buffers / of model_out
                    tb_lo, ver_lo = pick_text(texts[id(self.unet)])
                    hi2.set_text_version(ver)
                    hi2.forward_into(lat, float(t), tb[:2 * P], model_out[:2 * P])
                    lo1.set_text_version(ver_lo)
                    lo1.forward_into(lat, float(t), tb_lo[2 * P:], model_out[2 * P:])
                    n_dead_lo += 1
                else:
                    if tx["has_ver"]:
                        u.set_text_version(ver)
                    u.forward_into(lat if shared_latents else x_in, float(t), tb, model_out)
                if sld and g_rows is not None:
                    _lib.check(L.sdn_sld_guidance_rows(model_out.data_ptr(), P, D, g_rows.data_ptr(), sld["scale"], sld["thr"],
                                                       sld["ms"], sld["mb"], int(i >= sld["warmup"]), momentum.data_ptr(),
                                                       eps.data_ptr(), st), "sdn_sld_guidance_rows")
                elif sld:
                    _lib.check(L.sdn_sld_guidance(model_out.data_ptr(), P, D, float(guidance_scale), sld["scale"], sld["thr"],
                                                  sld["ms"], sld["mb"], int(i >= sld["warmup"]), momentum.data_ptr(),
                                                  eps.data_ptr(), st), "sdn_sld_guidance")
                elif g_rows is not None:
                    _lib.check(L.sdn_cfg_combine_rows(model_out.data_ptr(), P, nb, D, g_rows.data_ptr(), eps.data_ptr(), st),
                               "sdn_cfg_combine_rows")
                else:
                    _lib.check(L.sdn_cfg_combine(model_out.data_ptr(), P, nb, D, float(guidance_scale), eps.data_ptr(), st),
                               "sdn_cfg_combine")

                in_window = (kind is None) or (kind == "t" and lo <= t <= hi) or (kind == "i" and lo <= i <= hi)
                if in_window and repellency_processor is not None:
                    n_window += 1
                    sa, s1 = sch.sqrt_pair(t)
                    clip = sch.config.clip_sample_range if sch.config.clip_sample else 0.0
                    _lib.check(L.sdn_pred_x0(lat.data_ptr(), eps.data_ptr(), lat.numel(), sa, s1, clip, x0.data_ptr(), st),
                               "sdn_pred_x0")
                    if is_ddpm and t > 0:                       # scheduler.step() draws (and the caller discards) a randn
                        draw_discard()
                    src, isneg = self._condition(repellency_processor, x0, use_beta)
                    if self.record_den and self._last_den is not None:
                        den_log.append(self._last_den.clone())
                    if dev_flags:                               # no readback: the flag vector selects the drawing rows on the device
                        rng.draw_flagged(gens, noise, isneg)
                        fired_acc.add_(isneg.ne(0).to(torch.int32))
                        _lib.check(L.sdn_renoise_select(lat.data_ptr(), src.data_ptr(), noise.data_ptr(), isneg.data_ptr(),
                                                        P, D, sa, s1, st), "sdn_renoise_select")
                    else:
                        if use_flag:
                            flags = isneg.cpu().tolist()        # a readback: decides how many randn are drawn (tapes, host generators)
                        else:
                            flags = [1] * P
                            isneg = torch.ones(P, dtype=torch.int32, device=dev)
                        if any(flags):
                            fired = [p for p in range(P) if flags[p]]
                            draw_into(noise, fired if len(fired) < P else None, flags_dev=isneg)
                            n_renoise += len(fired)
                            _lib.check(L.sdn_renoise_select(lat.data_ptr(), src.data_ptr(), noise.data_ptr(), isneg.data_ptr(),
                                                            P, D, sa, s1, st), "sdn_renoise_select")

                co = sch.step_coefficients(t)
                z = None
                if is_ddpm and t > 0:
                    draw_into(noise)
                    z = noise
                clip = sch.config.clip_sample_range if sch.config.clip_sample else 0.0
                _lib.check(L.sdn_sched_step(lat.data_ptr(), eps.data_ptr(), None if z is None else z.data_ptr(),
                                            lat.numel(), co["sqrt_ac"], co["sqrt_1mac"], co["c_x0"], co["c_x"], co["c_eps"],
                                            co["sigma"] if z is not None else 0.0, clip, nxt.data_ptr(), st),
                           "sdn_sched_step")
                lat, nxt = nxt, lat
                if callback is not None and i % callback_steps == 0:
                    if dev_flags:
                        rng.sync_host()                         # a callback may look at the generators
                    callback(i, t, lat)
        finally:
            # undeclared again, whatever happened in the loop: a direct caller of the UNet gets plain forwards (and the handle
            # drops its cached text K / V), and the torch.Generator objects carry the offsets the device-side streams reached
            for u_ in nets:
                if texts[id(u_)]["has_ver"]:
                    u_.set_text_version(0)
            for sib in (hi2, lo1):
                if sib is not None:
                    sib.set_text_version(0)
            if dev_flags:
                rng.sync_host()
        if dev_flags:
            n_renoise = int(fired_acc.sum().item())                    # after the loop: the call's one count readback
        lat = lat.clone()                                              # the loop buffers are reused by the next call
        self.last_stats = {"renoise_draws": n_renoise, "window_steps": n_window, "prompts": P, "branches": nb,
                           "hi_steps": int(sum(use_hi)), "dead_branch_lo_steps": n_dead_lo, "window_readbacks": n_window if (use_flag and not dev_flags) else 0,
                           "tail_split": next((u_._tail_split_of(nb * P) for u_ in nets if hasattr(u_, "_tail_split_of")), None),
                           # how each UNet forward of this call ran: None = one launch plan, else [(first prompt, prompts, side stream)]
                           "forward_chunks": next((u_._chunks_of(nb * P) for u_ in nets if hasattr(u_, "_chunks_of")), None)}
        if self.record_den:
            self.last_stats["denominators"] = den_log
        if return_latents:
            return lat
        return self.decode_latents(lat, output_type)

    def _sibling(self, net, rep: int):
        """A second launch plan over `net`'s packed weights with another latent_repeat (kept for the pipeline's lifetime)."""
        from .unet import UNet2DConditionModel
        key = (id(net), rep)
        sib = self._sib.get(key)
        if sib is None:
            sib = self._sib[key] = UNet2DConditionModel(text_len=net.text_len, dtype=net.dtype, latent_repeat=rep,
                                                        precision=net.precision if net.precision in ("fp32", "bf16x3") else None,
                                                        **vars(net.config))
        sib._weights = net._weights                                   # the derived regions live in the buffer: nothing to prepare again
        return sib

    # ---- precision schedule ---------------------------------------------------------------------------------
    def hi_steps(self, timesteps, kind=None, lo=0, hi=0):
        """Which steps of the loop run on `unet_hi`.  `precision_schedule` is one of
          * a sequence of booleans, one per step;
          * a callable (i, t, in_window) -> bool;
          * a dict with any of: "first": K (the first K steps), "last": K (the last K steps), "window": True (every step
            inside the repellency window -- its gates are categorical decisions, ...threshold_time.py:552-569),
            "steps": an iterable of step indices.  The union of the parts is taken.
          * "all" / "none"."""
        n = len(timesteps)
        ps = self.precision_schedule

        def inw(i, t):
            return (kind is None) or (kind == "t" and lo <= t <= hi) or (kind == "i" and lo <= i <= hi)
        if ps is None or ps == "none":
            return [False] * n
        if ps == "all":
            return [True] * n
        if callable(ps):
            return [bool(ps(i, t, inw(i, t))) for i, t in enumerate(timesteps)]
        if isinstance(ps, dict):
            unknown = sorted(set(ps) - {"first", "last", "window", "steps"})
            if unknown:
                raise _lib.SdnError(f"precision_schedule: unknown keys {unknown}")
            on = set(int(k) for k in ps.get("steps", ()))
            on |= set(range(min(n, int(ps.get("first", 0)))))
            on |= set(range(max(0, n - int(ps.get("last", 0))), n))
            if ps.get("window"):
                on |= {i for i, t in enumerate(timesteps) if inw(i, t)}
            return [i in on for i in range(n)]
        seq = [bool(v) for v in ps]
        if len(seq) != n:
            raise _lib.SdnError(f"precision_schedule: {len(seq)} entries for {n} steps")
        return seq

    # ---- text front end (steps 3 of the reference's __call__) ---------------------------------------------
    def _tok(self, texts, padding="max_length", max_length=None, truncation=True):
        n = max_length or self.text_encoder.config.max_position_embeddings
        kw = dict(padding=padding, return_tensors="pt")
        if padding == "max_length":
            kw.update(max_length=n, truncation=truncation)
        t = self.tokenizer(texts, **kw)
        ids = t.input_ids if hasattr(t, "input_ids") else t["input_ids"]
        mask = getattr(t, "attention_mask", None) if hasattr(t, "input_ids") else t.get("attention_mask")
        if mask is None:               # a tokenizer that returns ids only: real tokens = everything up to the first end-of-text
            mask = (torch.arange(ids.shape[1])[None, :] <= ids.argmax(dim=-1, keepdim=True)).to(torch.int64)
        return ids, mask

    def _new_encode_prompt(self, prompt, negative_prompt=None, safety_concept: Optional[str] = None):
        """...threshold_time.py:231-349 for P prompts: tokenise with padding="max_length" / truncation, encode (no attention
        mask: the SD-v1.4 text encoder's config has no use_attention_mask), stack [unconditional | text] rows.
        With `safety_concept` (the SLD pipelines, modified_sld_pipeline_threshold_time.py:258-276) the concept text is encoded
        the same way and appended once per prompt: [unconditional | text | safety concept].
        Returns (embeddings [2P or 3P,77,768], input ids [P,77], the tokenizer's attention mask [P,77])."""
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        if negative_prompt is None:
            neg = [""] * len(prompts)
        elif isinstance(negative_prompt, str):
            neg = [negative_prompt] * len(prompts)
        else:
            neg = list(negative_prompt)
        if len(neg) != len(prompts):
            raise ValueError(f"`negative_prompt` has batch size {len(neg)}, but `prompt` has batch size {len(prompts)}")
        dev = torch.device("cuda", torch.cuda.current_device())
        ids, mask = self._tok(prompts)
        nids, _ = self._tok(neg)
        parts = [self.text_encoder(nids.to(dev))[0], self.text_encoder(ids.to(dev))[0]]
        if safety_concept is not None:
            cids, _ = self._tok([safety_concept])
            parts.append(self.text_encoder(cids.to(dev))[0].repeat(len(prompts), 1, 1))
        return torch.cat(parts), ids, mask

    def encode_prompt(self, prompt, negative_prompt=None) -> torch.Tensor:
        return self._new_encode_prompt(prompt, negative_prompt)[0]

    def _new_encode_negative_prompt_space(self, negative_prompt_space, max_length=77) -> torch.Tensor:
        """Pooled embeddings of the concept phrases (...threshold_time.py:186-209, pooler_output=True): encoded WITH the
        tokenizer's attention mask, as the reference does."""
        phrases = [negative_prompt_space] if isinstance(negative_prompt_space, str) else list(negative_prompt_space)
        dev = torch.device("cuda", torch.cuda.current_device())
        ids, mask = self._tok(phrases, max_length=max_length)
        return self.text_encoder(ids.to(dev), attention_mask=mask.to(dev)).pooler_output

    def _masked_ids(self, prompt: str) -> torch.Tensor:
        """Token ids of `_masked_encode_prompt` (...threshold_time.py:211-229): the prompt once per real token, that token
        replaced by id 0.  The reference feeds the UNPADDED [n, n + 2] rows; padding them to the encoder's 77 positions
        with the end-of-text id changes nothing it reads (causal attention; the pooled state is taken at the FIRST
        end-of-text token)."""
        ids, _ = self._tok([prompt], padding="longest")
        nmax = self.text_encoder.config.max_position_embeddings
        n_real = ids.shape[1] - 2
        if ids.shape[1] > nmax:
            ids = ids[:, :nmax]
            n_real = nmax - 2
        if n_real <= 0:
            return ids.new_zeros((0, nmax))
        rows = ids.repeat(n_real, 1)
        for i in range(n_real):
            rows[i, i + 1] = 0
        eot = int(ids.max())
        out = torch.full((n_real, nmax), eot, dtype=rows.dtype)
        out[:, :rows.shape[1]] = rows
        return out

    def _masked_encode_prompt(self, prompt: str) -> torch.Tensor:
        dev = torch.device("cuda", torch.cuda.current_device())
        return self.text_encoder(self._masked_ids(prompt).to(dev), attention_mask=None).pooler_output

    def _safree_prepare(self, prompt, text_embeddings: torch.Tensor, attn_mask: torch.Tensor, negative_prompt_space, sf):
        """The SAFREE block of the reference's __call__ (...threshold_time.py:458-486), once per prompt of the batch: concept
        projector from the pooled phrase embeddings, masked-token projector, token-wise replacement, and (svf) the number
        of leading steps that use the projected text.  fp32 torch ops on the GPU (once per prompt, outside the step loop).
        Returns rescaled_text_embeddings [2P,77,768] ([uncond | projected text] rows) and per-prompt beta_adjusted."""
        from . import safree
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        P = len(prompts)
        E = text_embeddings.float()
        negspace = self._new_encode_negative_prompt_space(negative_prompt_space, 77).float()
        P_c = safree.projection_matrix(negspace.T)
        # one text-encoder call for the masked variants of every prompt
        dev = E.device
        rows = [self._masked_ids(p_) for p_ in prompts]
        counts = [r.shape[0] for r in rows]
        allrows = torch.cat(rows).to(dev)
        pooled = []
        for lo in range(0, allrows.shape[0], 256):
            pooled.append(self.text_encoder(allrows[lo:lo + 256], attention_mask=None).pooler_output.float())
        pooled = torch.cat(pooled) if pooled else E.new_zeros((0, E.shape[-1]))
        masked_list, lo = [], 0
        for p_ in range(P):
            masked_list.append(pooled[lo:lo + counts[p_]])
            lo += counts[p_]
        if self.batched_safree and min(counts) > 0:
            r = safree.prepare_batch(E, masked_list, negspace, attn_mask, alpha=sf["alpha"], svf=bool(sf["svf"]), up_t=sf["up_t"],
                                     category=sf["category"], concept_proj=P_c)
            resc_rows = list(r["rescaled_text_embeddings"][P:])
            betas, adjusted, removed, masks = r["beta"], r["beta_adjusted"], r["n_removed"], list(r["token_mask"])
        else:
            resc_rows, betas, adjusted, removed, masks = [], [], [], [], []
            for p_ in range(P):
                pair = torch.stack([E[p_], E[P + p_]])
                r = safree.prepare(pair, masked_list[p_], negspace, attn_mask[p_].to(dev), alpha=sf["alpha"], svf=bool(sf["svf"]),
                                   up_t=sf["up_t"], category=sf["category"], concept_proj=P_c)
                resc_rows.append(r["rescaled_text_embeddings"][1])
                betas.append(r["beta"]); adjusted.append(r["beta_adjusted"]); removed.append(r["n_removed"]); masks.append(r["token_mask"])
        log = sf.get("logger")
        if log is not None:
            for p_ in range(P):
                log.log(f"Among {counts[p_]} tokens, we remove {removed[p_]}.")
                if sf["svf"]:
                    log.log(f"beta : {betas[p_]}, adjusted_beta: {adjusted[p_]}")
        rescaled = torch.cat([E[:P], torch.stack(resc_rows)])
        return {"rescaled_text_embeddings": rescaled, "beta_adjusted": adjusted if sf["svf"] else None, "beta": betas,
                "n_removed": removed, "negspace": negspace, "token_mask": torch.stack(masks)}

    def decode_latents(self, latents: torch.Tensor, output_type: str = "np"):
        """Steps 8-10 of the reference's __call__ (...threshold_time.py:588-596)."""
        if output_type == "uint8":
            return self.vae.decode_latents_uint8(latents)
        image = self.vae.decode_latents(latents)                      # NHWC float32 numpy in [0, 1]
        if output_type == "pil":
            from PIL import Image                                     # numpy_to_pil
            return [Image.fromarray(im) for im in (image * 255).round().astype("uint8")]
        return image

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _branches(first: torch.Tensor, base: torch.Tensor, nb: int) -> torch.Tensor:
        if nb == 2:
            return first
        _, text_e = base.chunk(2)
        return torch.cat([first, text_e])

    @staticmethod
    def _generators(generator, P: int, dev) -> Sequence[torch.Generator]:
        if generator is None:
            return [torch.Generator(device=dev).manual_seed(1000 + p) for p in range(P)]
        if isinstance(generator, torch.Generator):
            if P != 1:
                raise _lib.SdnError("pass a list of P generators (one per prompt) when batching prompts")
            return [generator]
        if len(generator) != P:
            raise _lib.SdnError(f"need {P} generators, got {len(generator)}")
        return list(generator)

    def _condition(self, proc, x0: torch.Tensor, use_beta: bool):
        """Device-side conditioning.  Returns (tensor to re-noise from, is_negation int32[P])."""
        from .repellency import repellency_methods_threshold as thr
        if hasattr(proc, "conditioning_device"):
            returns_neg = isinstance(proc, thr.RBFKernelRepellency) and not use_beta   # conditioning_1 quirk, :190-193
            neg, den, isneg = proc.conditioning_device(x0, beta_threshold=use_beta, want_neg=returns_neg)
            self._last_den = den
            return (neg if returns_neg else x0), isneg
        out = proc.conditioning(x0, beta_threshold=use_beta)                            # generic plug-in (host flags)
        flag = out.get("is_negation", False)
        isneg = torch.full((x0.shape[0],), int(bool(flag)), dtype=torch.int32, device=x0.device)
        return out["x_0_hat"].contiguous(), isneg


def make_scheduler(name: str = "ddpm"):
    """"ddpm" = the live SD-v1.4 scheduler of the reference (run_nudity.py:108); "ddim" = the one BASELINE names."""
    return DDPMScheduler() if name.lower() == "ddpm" else DDIMScheduler()
