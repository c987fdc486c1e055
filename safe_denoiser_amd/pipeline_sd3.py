"""SD-v3 denoising loop with flow-matching repellency re-noise -- batched over prompts.

Restates the hot loop of models/sdv3/safe_denoiser_pipeline.py:1105-1165 on libsdn kernels:
    v = transformer(cat([lat]*2), t, cat([neg_embeds, embeds]), cat([neg_pooled, pooled]))      :1112-1127
    v = v_u + g (v_t - v_u)                                                                     :1130-1132
    780 <= t <= 1000 (with a processor):
        x0 = lat - sigma v ; x1 = lat + (1 - sigma) v                   (sdn_flow_endpoints)    :1142-1148
        x0r = processor.conditioning(x0, beta_threshold=False)["x_0_hat"] (fast_sdv3: channel-normalised query) :1150-1152
        lat = x0r + sigma_next (sqrt(sigma_next) x1 + sqrt(1 - sigma_next) z - x0r)   (sdn_flow_renoise) :1159-1160
    else: lat = scheduler.step(v, t, lat)       (Euler)                                          :1165
    lat = lat.to(latents_dtype)   (the reference keeps fp16 latents between steps)               :1161,1167-1171
Text encoders (CLIP x2 + T5) are outside the hot path: pass `prompt_embeds` [2P,333,4096] and
`pooled_prompt_embeds` [2P,2048] ([P negative | P positive]) -- or the reference's four separate tensors (`prompt_embeds`,
`negative_prompt_embeds`, `pooled_prompt_embeds`, `negative_pooled_prompt_embeds`, each [P, ...]; concatenated as :1069,1081-1082) --
or, round 5, PROMPT STRINGS as the reference's call takes them (:862-891, run_nudity_sdv3.py:351-360) with a caller-supplied
`text_front_end` (constructor argument) that owns the three third-party encoders and exposes the three calls the reference makes:
`encode_prompt(prompt=, negative_prompt=, ...)` -> (prompt_embeds, negative_prompt_embeds, pooled, negative_pooled) (:1038-1059),
`masked_encode_prompt(prompt)` -> [n_tokens, 4096] (:773-860) and `encode_negative_prompt_space(phrases)` -> [n_phrases, 4096]
(:722-771).  The orchestration between them is the reference's: the negative prompt is OVERWRITTEN with the 17 joined concept
phrases (:988-996), the SAFREE projection (`mask_to_onp`, :1061-1078) runs per prompt, the projected text is fed at every step.  `return_latents=True` (the engine-side default: the loop's own
output) gives the latents; with a `vae` and `return_latents=False` the call ends like the reference's (:1195-1214) and returns
`StableDiffusion3PipelineOutput(images=...)` -- `.images` as run_nudity_sdv3.py:351-360 reads it (`return_dict=False`: a tuple;
`output_type="latent"`: the latents in `.images`).
The reference draws z from the GLOBAL torch RNG (`randn_like`, :1159); so does this loop unless `noise_fn` is given -- the P draws
of a window step come out of one launch (rng.BatchedNormal.draw_sequence: the same values and the same final generator offset as
P consecutive `torch.randn` calls on the default CUDA generator).  The projection runs through `conditioning_device` (no host
synchronisation: the plug-in's dict API reads the score back with .item()), and the loop's tensors are kept across calls.
(No `latent_repeat` here: the two guidance branches of the MMDiT share nothing but the patch embedding -- their text enters the
first joint-attention block -- so there is no branch-independent prefix to compute once.)
Known reference quirk kept out: with `repellency_processor=None` inside the window the reference reads undefined
names (:1159); here that case simply takes the Euler step.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch

from . import _lib


class StableDiffusion3PipelineOutput:
    """What the reference's call returns (`pipe(...).images`, run_nudity_sdv3.py:351-360; safe_denoiser_pipeline.py:1214)."""
    __slots__ = ("images",)

    def __init__(self, images):
        self.images = images


# the concept phrases hard-coded in the reference's __call__ (models/sdv3/safe_denoiser_pipeline.py:988-994)
SD3_NEGATIVE_PROMPT_SPACE = ["Sexual Acts", "Content Meant to Arouse Sexual Excitement", "Nudity", "Pornography", "Erotic Art", "Lustful",
                             "Seductive", "Orgasmic", "Libido", "Kinky", "Sexual Orientation", "Sexual Attraction", "Sexual Intercourse",
                             "Sexual Pleasure", "Sexual Fantasy", "Carnal Desires", "Sexual Gratification"]


class SD3SafeDenoiserPipeline:
    def __init__(self, transformer, scheduler, vae=None, text_front_end=None):
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae
        self.text_front_end = text_front_end   # the caller's CLIP x2 + T5 stack (see the module docstring); None = embeddings only
        self.vae_scale_factor = 8
        self.last_stats = {}
        self._bufs = {}
        self._rng = {}
        self.batched_rng = True     # False: one torch.randn per prompt, as the reference's Python loop would draw

    @torch.no_grad()
    def __call__(self, prompt=None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 50, guidance_scale: float = 7.0, generator=None, latents=None,
                 prompt_embeds: Optional[torch.Tensor] = None, pooled_prompt_embeds: Optional[torch.Tensor] = None,
                 repellency_processor=None, latents_dtype=torch.float16, return_latents: bool = True,
                 noise_fn: Optional[Callable] = None, rescaled_text_embeddings: Optional[torch.Tensor] = None,
                 masked_embs=None, negspace_embs: Optional[torch.Tensor] = None, safree_alpha: float = 0.01,
                 negative_prompt=None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 negative_pooled_prompt_embeds: Optional[torch.Tensor] = None, **kwargs):
        _lib.require_gpu()
        if prompt_embeds is None and prompt is not None and self.text_front_end is not None:
            # the reference's steps 1-3 (:985-1078) for P prompts: its own negative prompt, the three encoder calls, and the inputs
            # of the SAFREE projection (which runs below, per prompt, on the GPU)
            fe = self.text_front_end
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            neg = ", ".join(SD3_NEGATIVE_PROMPT_SPACE)                        # (:996: whatever the caller passed is overwritten)
            pe, npe, pp, npp = fe.encode_prompt(prompt=prompts, negative_prompt=[neg] * len(prompts))
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = pe, npe, pp, npp
            if masked_embs is None and rescaled_text_embeddings is None:
                masked_embs = [fe.masked_encode_prompt(p_) for p_ in prompts]
                negspace_embs = fe.encode_negative_prompt_space(SD3_NEGATIVE_PROMPT_SPACE)
        if negative_prompt_embeds is not None:                                 # the reference's four-tensor form -> [negative | positive]
            if negative_pooled_prompt_embeds is None or prompt_embeds is None or pooled_prompt_embeds is None:
                raise _lib.SdnError("negative_prompt_embeds needs prompt_embeds, pooled_prompt_embeds and negative_pooled_prompt_embeds")
            if negative_prompt_embeds.shape != prompt_embeds.shape or negative_pooled_prompt_embeds.shape != pooled_prompt_embeds.shape:
                raise _lib.SdnError("negative / positive embeddings must have the same shapes")
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds], dim=0)                     # (:1069,1081)
            pooled_prompt_embeds = torch.cat([negative_pooled_prompt_embeds, pooled_prompt_embeds], dim=0)   # (:1082)
        if prompt_embeds is None or pooled_prompt_embeds is None:
            raise NotImplementedError("text encoders are outside the hot path: pass prompt_embeds [2P,T,4096] and "
                                      "pooled_prompt_embeds [2P,2048] ([P negative | P positive]), or construct the pipeline with "
                                      "text_front_end= (the caller's CLIP x2 + T5 stack) and pass `prompt` strings")
        output_type = kwargs.get("output_type", "pil")
        return_dict = kwargs.get("return_dict", True)
        if output_type == "latent":                                          # the reference's own latent tap (:1195-1196)
            return_latents = True
        if not return_latents:
            if self.vae is None:
                raise NotImplementedError("no VAE attached: construct with vae=AutoencoderKL(**SD3_VAE_CONFIG) or use return_latents=True")
            if output_type not in ("pil", "np", "uint8"):
                raise _lib.SdnError("output_type must be 'pil', 'np', 'uint8' or 'latent'")
        hi = kwargs.get("negation_warmup_start", 1000)
        lo = kwargs.get("negation_warmup_end", 780)
        dev = torch.device("cuda", torch.cuda.current_device())
        tr = self.transformer
        P = prompt_embeds.shape[0] // 2
        s, C_ = tr.config.sample_size, tr.config.in_channels
        if height is not None and (height // self.vae_scale_factor, (width or height) // self.vae_scale_factor) != (s, s):
            raise _lib.SdnError(f"this MMDiT plan is built for {s * 8}x{s * 8} images")
        shape1 = (1, C_, s, s)
        # SAFREE on the T5-side embeddings (models/sdv3/safe_denoiser_pipeline.py:1061-1078): the reference's loop feeds the
        # projected text to the transformer at EVERY step (:1115).  The T5 / CLIP encoders are outside this engine: the caller
        # passes either the finished `rescaled_text_embeddings` [2P,T,4096], or the first-token T5 states it takes its
        # projectors from (`masked_embs`: one [n_tokens, 4096] tensor per prompt, `negspace_embs` [n_phrases, 4096]).
        if rescaled_text_embeddings is None and masked_embs is not None and negspace_embs is not None:
            from . import safree
            E = prompt_embeds.to(dev)
            rows = []
            for p_ in range(P):
                pair = torch.stack([E[p_], E[P + p_]])
                rows.append(safree.prepare_sd3(pair, masked_embs[p_].to(dev), negspace_embs.to(dev), alpha=safree_alpha)
                            ["rescaled_text_embeddings"][1])
            rescaled_text_embeddings = torch.cat([E[:P], torch.stack(rows).to(E.dtype)])
        text_src = prompt_embeds if rescaled_text_embeddings is None else rescaled_text_embeddings
        text = tr.prepare_text(text_src.to(dev))
        pooled = pooled_prompt_embeds.to(device=dev, dtype=tr.dtype).contiguous()

        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        ts = [float(t) for t in sch._ts_host]
        sig_step = [float(x) for x in sch._sig_host]                         # Euler grid (sigma_n = 0 appended)

        def rq(x):                                                            # latents.to(latents_dtype) round trip
            return x if latents_dtype == torch.float32 else x.to(latents_dtype).float()

        def draw(p):
            if noise_fn is not None:
                return noise_fn(p, shape1).to(device=dev, dtype=torch.float32)
            return torch.randn(shape1, device=dev, dtype=latents_dtype).float()     # global RNG, as randn_like (:1159)

        if latents is None:
            if noise_fn is not None:
                lat = torch.cat([draw(p) for p in range(P)])
            else:
                gens = generator if isinstance(generator, (list, tuple)) else [generator] * P
                lat = torch.cat([torch.randn(shape1, generator=gens[p], device=dev, dtype=latents_dtype).float()
                                 for p in range(P)])
        else:
            lat = latents.to(device=dev, dtype=torch.float32).clone()
        lat = rq(lat).contiguous()

        L, st = _lib.lib(), _lib.stream_ptr()
        D = C_ * s * s
        key = (P, C_, s, str(dev))
        if self._bufs.get("key") != key:                                       # persistent across calls (and across steps)
            f32 = dict(dtype=torch.float32, device=dev)
            self._bufs = dict(key=key, x_in=torch.empty((2 * P, C_, s, s), **f32), vout=torch.empty((2 * P, C_, s, s), **f32),
                              **{n: torch.empty((P, C_, s, s), **f32) for n in ("v", "x0", "x1", "z", "a", "b")})
        bf = self._bufs
        x_in, vout, v, x0, x1, z = (bf[n] for n in ("x_in", "vout", "v", "x0", "x1", "z"))
        cur, nxt = bf["a"], bf["b"]
        cur.copy_(lat)
        rng = None
        if noise_fn is None and self.batched_rng:
            from .rng import BatchedNormal
            rng = self._rng.get((str(dev), D))
            if rng is None:
                rng = self._rng[(str(dev), D)] = BatchedNormal(dev, D)
        device_proj = hasattr(repellency_processor, "conditioning_device")

        def rq_(x):                                                            # in place: the latents_dtype round trip of a loop buffer
            if latents_dtype != torch.float32:
                x.copy_(x.to(latents_dtype))
            return x

        n_win = 0
        for i, t in enumerate(ts):
            x_in.view(2, P, C_, s, s).copy_(cur)
            tr.forward_into(x_in, t, text, pooled, vout)
            vq = vout if latents_dtype == torch.float32 else vout.to(latents_dtype).float()   # model output is fp16 in the ref
            _lib.check(L.sdn_cfg_combine(vq.data_ptr(), P, 2, D, float(guidance_scale), v.data_ptr(), st), "sdn_cfg_combine")
            if lo <= t <= hi and repellency_processor is not None:
                n_win += 1
                sigma = t / 1000.0
                sigma_next = ts[i + 1] / 1000.0 if i + 1 < len(ts) else 0.0
                _lib.check(L.sdn_flow_endpoints(cur.data_ptr(), v.data_ptr(), cur.numel(), sigma, x0.data_ptr(),
                                                x1.data_ptr(), st), "sdn_flow_endpoints")
                rq_(x0); rq_(x1)
                if device_proj:                                                # fast_sdv3: x0 <- x0 - scale * neg, in place, no sync
                    repellency_processor.conditioning_device(x0, beta_threshold=False, want_neg=False)
                    x0r = x0
                else:
                    x0r = repellency_processor.conditioning(x0, beta_threshold=False)["x_0_hat"].float().contiguous()
                if rng is not None:                                            # the global generator's next P draws, one launch
                    rng.draw_sequence(torch.cuda.default_generators[dev.index], z, shape1)
                    rq_(z)                                                     # (randn_like of an fp16 tensor: the f32 normal, rounded)
                else:
                    for p in range(P):
                        z[p:p + 1] = draw(p)
                _lib.check(L.sdn_flow_renoise(x0r.data_ptr(), x1.data_ptr(), z.data_ptr(), cur.numel(), sigma_next,
                                              nxt.data_ptr(), st), "sdn_flow_renoise")
            else:
                _lib.check(L.sdn_flow_euler_step(cur.data_ptr(), v.data_ptr(), cur.numel(), sig_step[i], sig_step[i + 1],
                                                 nxt.data_ptr(), st), "sdn_flow_euler_step")
            rq_(nxt)
            cur, nxt = nxt, cur
        lat = cur.clone()                                                       # the loop buffers are reused by the next call
        self.last_stats = {"window_steps": n_win, "prompts": P}
        wrap = (lambda im: StableDiffusion3PipelineOutput(im) if return_dict else (im,))
        if return_latents:
            return wrap(lat.to(latents_dtype)) if output_type == "latent" else lat.to(latents_dtype)
        # safe_denoiser_pipeline.py:1195-1199: latents / scaling_factor + shift_factor -> vae.decode -> postprocess
        lat = lat.to(latents_dtype).float()
        if output_type == "uint8":
            return wrap(self.vae.decode_latents_uint8(lat))
        image = self.vae.decode_latents(lat)
        if output_type == "pil":
            from PIL import Image
            image = [Image.fromarray(im) for im in (image * 255).round().astype("uint8")]
        return wrap(image)
