"""The denoising loop with per-step repellency projection -- batched over prompts.

Call surface: the reference's `ModifiedSafreeDiffusionPipeline_Rep.__call__`
(models/textuals_visual/modified_safree_diffusion_pipeline_threshold_time.py:352-375,598) restricted to the hot path:
the tensor-level inputs the loop consumes (`prompt_embeddings`, `generator`, `latents`, `repellency_processor`,
`safree_dict`, `negation_warmup_start/end`, `return_latents`, `output_type`).  The text encoder sits before the loop
 and is not part of this engine: pass CLIP hidden states as `prompt_embeddings`
([2P,77,768] = chunk(2) -> [P uncond | P text], or [3P,...] with `lra`).  With a `vae` (safe_denoiser_amd.vae.
AutoencoderKL, section 8f row 2) and `return_latents=False` the call ends like the reference's (:588-596):
`decode_latents` -> NHWC float numpy in [0,1] (`output_type="np"`), PIL images ("pil", needs Pillow) or the uint8
NHWC tensor numpy_to_pil would build, left on the device ("uint8"); otherwise latents come back.

What differs from the reference by design (results per sample are the same):
  * P prompts are denoised together (the reference is hard-wired to batch 1); every prompt keeps its OWN
    torch.Generator so its random stream -- latents, the discarded variance draw of the x0 probe, the conditional
    re-noise draw, the step's variance draw, in that order (row S2) -- is exactly the reference's;
  * all per-element math is in libsdn kernels; the only host<->device traffic in the loop is ONE readback of the
    P is_negation flags per repellency-window step (the reference syncs three times per step per prompt), needed
    only because the number of randn draws depends on the flag.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

from . import _lib
from .schedulers import DDIMScheduler, DDPMScheduler

# gating variants of the reference's pipeline files: (window kind, lo, hi,
# beta_threshold kwarg, honours is_negation)
VARIANTS = {
    "threshold_time": ("t", 780, 1000, True, True),     # *_threshold_time.py  (north-star path)
    "time": ("t", 800, 1000, False, False),             # *_time.py
    "threshold": ("i", 0, 50, True, True),              # *_threshold.py
    "plain": (None, 0, 0, False, True),                 # modified_{safree,sld}_diffusion_pipeline.py: every step
}


class SafeDenoiserPipeline:
    GRAPH_MAX_BATCH = 16      # UNet rows (branches x prompts) up to which a forward is launch-bound and replayed as a hipGraph

    def __init__(self, unet, scheduler, variant: str = "threshold_time", vae=None, use_graphs: Optional[bool] = None,
                 text_encoder=None, tokenizer=None, split_k: bool = False):
        if variant not in VARIANTS:
            raise KeyError(f"unknown variant {variant}; have {sorted(VARIANTS)}")
        self.unet, self.scheduler, self.variant, self.vae = unet, scheduler, variant, vae
        self.use_graphs = use_graphs          # None = automatic: on for small batches (the reference's one-prompt calls)
        # optional front end: safe_denoiser_amd.clip.CLIPTextModel + the caller's CLIPTokenizer
        self.text_encoder, self.tokenizer = text_encoder, tokenizer
        # split_k=True: small batches additionally run their long, thin GEMMs in split-K form (lower single-prompt latency;
        # a prompt's result then depends, in the last fp32 bits, on how many prompts share its batch)
        self.split_k = split_k
        self.vae_scale_factor = 8
        self.last_stats = {}

    # ------------------------------------------------------------------------------------------------
    def _noise(self, noise_fn, generators, p: int, shape, device):
        if noise_fn is not None:
            return noise_fn(p, shape).to(device=device, dtype=torch.float32)
        return torch.randn(shape, generator=generators[p], device=device, dtype=torch.float32)

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, generator=None, latents=None,
                 prompt_embeddings: Optional[torch.Tensor] = None, repellency_processor=None, safree_dict=None,
                 rescaled_text_embeddings: Optional[torch.Tensor] = None, beta_adjusted: Optional[int] = None,
                 return_latents: bool = True, noise_fn: Optional[Callable] = None, output_type: str = "pil", **kwargs):
        _lib.require_gpu()
        if prompt_embeddings is None:
            if prompt is None or self.text_encoder is None or self.tokenizer is None:
                raise NotImplementedError("pass `prompt_embeddings` ([2P,77,768]), or construct the pipeline with text_encoder= "
                                          "(safe_denoiser_amd.clip.CLIPTextModel) and tokenizer= and pass `prompt` strings")
            prompt_embeddings = self.encode_prompt(prompt, kwargs.get("negative_prompt"))
        if not return_latents:
            if self.vae is None:
                raise NotImplementedError("no VAE decoder attached: construct the pipeline with vae=AutoencoderKL(...) or "
                                          "use return_latents=True (the reference's parity tap, ...threshold_time.py:585-586)")
            if output_type not in ("pil", "np", "uint8"):
                raise _lib.SdnError("output_type must be 'pil', 'np' or 'uint8'")
        sf = dict(safree=False, svf=False, lra=False, re_attn_t=(-1, -1))
        if safree_dict:
            sf.update(safree_dict)
        # SLD family (modified_sld_pipeline*.py): third branch = safety concept, guidance eq. 3-8 with momentum state
        sld = None
        if kwargs.get("sld_guidance_scale", 0) and kwargs["sld_guidance_scale"] >= 1:
            sld = dict(scale=float(kwargs["sld_guidance_scale"]), warmup=int(kwargs.get("sld_warmup_steps", 10)),
                       thr=float(kwargs.get("sld_threshold", 0.01)), ms=float(kwargs.get("sld_momentum_scale", 0.3)),
                       mb=float(kwargs.get("sld_mom_beta", 0.4)))
        kind, lo_default, hi_default, use_beta, use_flag = VARIANTS[self.variant]
        hi = kwargs.get("negation_warmup_start", hi_default)          # reference: t <= start and t >= end
        lo = kwargs.get("negation_warmup_end", lo_default)
        nb = 3 if (sf["lra"] or sld) else 2
        if guidance_scale <= 1.0:
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
        tb_safe = self.unet.prepare_text(E_safe) if E_safe is not None else None

        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        timesteps = [int(t) for t in sch.timesteps]

        gens = self._generators(generator, P, dev) if noise_fn is None else None
        if latents is None:
            lat = torch.empty((P, C_, s, s), dtype=torch.float32, device=dev)
            for p in range(P):
                lat[p:p + 1] = self._noise(noise_fn, gens, p, shape1, dev)
            lat = lat * sch.init_noise_sigma if sch.init_noise_sigma != 1.0 else lat
        else:
            lat = latents.to(device=dev, dtype=torch.float32).clone()

        L = _lib.lib()
        st = _lib.stream_ptr()
        # a UNet built with latent_repeat = nb repeats the latents itself (and shares the branch-independent prefix)
        rep = getattr(self.unet, "latent_repeat", 1)
        if rep not in (1, nb):
            raise _lib.SdnError(f"unet.latent_repeat = {rep} but this call runs {nb} guidance branches")
        shared_latents = rep == nb
        if hasattr(self.unet, "set_graph_mode"):
            small = nb * P <= self.GRAPH_MAX_BATCH if self.use_graphs is None else bool(self.use_graphs)
            self.unet.set_graph_mode(small)
            if hasattr(self.unet, "set_split_k"):
                self.unet.set_split_k(bool(self.split_k) and small)
        x_in = None if shared_latents else torch.empty((nb * P, C_, s, s), dtype=torch.float32, device=dev)
        model_out = torch.empty((nb * P, C_, s, s), dtype=torch.float32, device=dev)
        eps = torch.empty((P, C_, s, s), dtype=torch.float32, device=dev)
        x0 = torch.empty_like(eps)
        noise = torch.empty_like(eps)
        nxt = torch.empty_like(eps)
        is_ddpm = isinstance(sch, DDPMScheduler)
        n_renoise = 0
        n_window = 0
        momentum = torch.zeros_like(eps) if sld else None

        for i, t in enumerate(timesteps):
            if not shared_latents:
                x_in.view(nb, P, C_, s, s).copy_(lat)                               # cat([latents] * nb)
            if sf["svf"]:
                use_safe = tb_safe is not None and beta_adjusted is not None and i <= beta_adjusted
            else:
                use_safe = tb_safe is not None and sf["re_attn_t"][0] <= i <= sf["re_attn_t"][1]
            self.unet.forward_into(lat if shared_latents else x_in, float(t), tb_safe if use_safe else tb_plain, model_out)
            if sld:
                _lib.check(L.sdn_sld_guidance(model_out.data_ptr(), P, D, float(guidance_scale), sld["scale"], sld["thr"],
                                              sld["ms"], sld["mb"], int(i >= sld["warmup"]), momentum.data_ptr(),
                                              eps.data_ptr(), st), "sdn_sld_guidance")
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
                    for p in range(P):
                        self._noise(noise_fn, gens, p, shape1, dev)
                src, isneg = self._condition(repellency_processor, x0, use_beta)
                if use_flag:
                    flags = isneg.cpu().tolist()            # the one readback: decides how many randn are drawn
                else:
                    flags = [1] * P
                    isneg = torch.ones(P, dtype=torch.int32, device=dev)
                if any(flags):
                    for p in range(P):
                        if flags[p]:
                            noise[p:p + 1] = self._noise(noise_fn, gens, p, shape1, dev)
                            n_renoise += 1
                    _lib.check(L.sdn_renoise_select(lat.data_ptr(), src.data_ptr(), noise.data_ptr(), isneg.data_ptr(),
                                                    P, D, sa, s1, st), "sdn_renoise_select")

            co = sch.step_coefficients(t)
            z = None
            if is_ddpm and t > 0:
                for p in range(P):
                    noise[p:p + 1] = self._noise(noise_fn, gens, p, shape1, dev)
                z = noise
            clip = sch.config.clip_sample_range if sch.config.clip_sample else 0.0
            _lib.check(L.sdn_sched_step(lat.data_ptr(), eps.data_ptr(), None if z is None else z.data_ptr(),
                                        lat.numel(), co["sqrt_ac"], co["sqrt_1mac"], co["c_x0"], co["c_x"], co["c_eps"],
                                        co["sigma"] if z is not None else 0.0, clip, nxt.data_ptr(), st),
                       "sdn_sched_step")
            lat, nxt = nxt, lat

        self.last_stats = {"renoise_draws": n_renoise, "window_steps": n_window, "prompts": P, "branches": nb}
        if return_latents:
            return lat
        return self.decode_latents(lat, output_type)

    def encode_prompt(self, prompt, negative_prompt=None) -> torch.Tensor:
        """_encode_prompt of the reference (...threshold_time.py:262-349) without the SAFREE token masking: tokenise with
        padding="max_length" / truncation, encode, and stack [unconditional | text] rows for classifier-free guidance."""
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        neg = [""] * len(prompts) if negative_prompt is None else (
            [negative_prompt] * len(prompts) if isinstance(negative_prompt, str) else list(negative_prompt))
        if len(neg) != len(prompts):
            raise _lib.SdnError("negative_prompt must match the number of prompts")
        n = self.text_encoder.config.max_position_embeddings
        dev = torch.device("cuda", torch.cuda.current_device())

        def ids_of(texts):
            t = self.tokenizer(texts, padding="max_length", max_length=n, truncation=True, return_tensors="pt")
            return (t.input_ids if hasattr(t, "input_ids") else t["input_ids"]).to(dev)

        return torch.cat([self.text_encoder(ids_of(neg))[0], self.text_encoder(ids_of(prompts))[0]])

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

    @staticmethod
    def _condition(proc, x0: torch.Tensor, use_beta: bool):
        """Device-side conditioning.  Returns (tensor to re-noise from, is_negation int32[P])."""
        from .repellency import repellency_methods_threshold as thr
        if hasattr(proc, "conditioning_device"):
            returns_neg = isinstance(proc, thr.RBFKernelRepellency) and not use_beta   # conditioning_1 quirk, :190-193
            neg, _den, isneg = proc.conditioning_device(x0, beta_threshold=use_beta, want_neg=returns_neg)
            return (neg if returns_neg else x0), isneg
        out = proc.conditioning(x0, beta_threshold=use_beta)                            # generic plug-in (host flags)
        flag = out.get("is_negation", False)
        isneg = torch.full((x0.shape[0],), int(bool(flag)), dtype=torch.int32, device=x0.device)
        return out["x_0_hat"].contiguous(), isneg


def make_scheduler(name: str = "ddpm"):
    """"ddpm" = the live SD-v1.4 scheduler of the reference (run_nudity.py:108); "ddim" = the one BASELINE names."""
    return DDPMScheduler() if name.lower() == "ddpm" else DDIMScheduler()
