"""UNet2DConditionModel front-end: the call surface the reference's pipelines use
(`self.unet(latent_model_input, t, encoder_hidden_states=E).sample`, ...threshold_time.py:538,540;
`self.unet.config.sample_size`, `self.unet.in_channels`), executed by libsdn's static launch plan.

Weights: a diffusers-keyed state_dict (the names of `UNet2DConditionModel.state_dict()` for SD-v1.4) is packed
ONCE into a single device buffer in the engine's layouts, driven by the manifest the C side publishes
(sdn_unet_param_info): conv kernels [O,I,3,3] -> [O][ky][kx][I] bf16, linears / 1x1 convs -> [O][I] bf16,
GEGLU projection rows interleaved value/gate in blocks of 16, norms and biases f32.
dtype=torch.float32 selects the plan's PRECISION mode (sdn_unet_config.dtype 2): weights, text and activations stay f32 and
the contractions run on the f32-input matrix cores -- 1/16 of the 16-bit rate, ~1e-6 per forward from the fp32 reference.
"""
from __future__ import annotations

import ctypes as C
import math
from types import SimpleNamespace

import torch

from . import _lib

SD14_CONFIG = dict(in_channels=4, out_channels=4, sample_size=64, block_out_channels=(320, 640, 1280, 1280),
                   down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D",
                                     "DownBlock2D"),
                   layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32)

P_VEC_F32, P_MAT, P_CONV3X3, P_GEGLU_MAT, P_GEGLU_VEC = 0, 1, 2, 3, 4
P_DERIVED = 6      # regions the engine fills itself (sdn_unet_prepare): not state_dict tensors


class UNetOutput:
    __slots__ = ("sample",)

    def __init__(self, sample):
        self.sample = sample


def _interleave16(t: torch.Tensor) -> torch.Tensor:
    """[2F, ...] (value rows then gate rows) -> blocks of 16 value rows followed by their 16 gate rows."""
    f = t.shape[0] // 2
    v, g = t[:f], t[f:]
    rest = t.shape[1:]
    return torch.stack([v.reshape(f // 16, 16, *rest), g.reshape(f // 16, 16, *rest)], dim=1).reshape(2 * f, *rest)


class UNet2DConditionModel:
    def __init__(self, text_len: int = 77, dtype=torch.bfloat16, latent_repeat: int = 1, precision: str | None = None, **config):
        """precision = "bf16x3" (with dtype=torch.float32, or alone): fp32 storage with split-operand contractions on the bf16
        matrix cores (sdn_unet_config.dtype 3, sdn_gemm_x3) -- the mode that meets the north star's 1e-3 at a multiple of the
        f32-MFMA plan's speed.
        latent_repeat = r > 1: the engine-side form of `torch.cat([latents] * r)` (classifier-free guidance): `sample`
        then holds B / r latents, `encoder_hidden_states` stays [B] branch-major, and everything up to the first
        cross-attention is computed once per latent (sdn_unet_config.latent_repeat).  Bit-identical results."""
        if precision not in (None, "fp32", "bf16x3"):
            raise _lib.SdnError('precision must be None, "fp32" or "bf16x3"')
        if precision is not None:
            dtype = torch.float32                                   # both precision modes store f32
        if dtype not in (torch.bfloat16, torch.float16, torch.float32):
            raise _lib.SdnError("storage dtype must be torch.bfloat16, torch.float16 or torch.float32 (the precision mode)")
        self.dtype = dtype
        self.precision = precision or ("fp32" if dtype == torch.float32 else None)
        self.latent_repeat = max(1, int(latent_repeat))
        cfg = dict(SD14_CONFIG)
        cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        self.in_channels = cfg["in_channels"]
        self.text_len = text_len
        boc = list(cfg["block_out_channels"])
        n = len(boc)
        c = _lib.UnetConfig(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
                            sample_size=cfg["sample_size"], n_levels=n,
                            block_out_channels=(C.c_int32 * 4)(*(boc + [0] * (4 - n))),
                            level_has_attn=(C.c_int32 * 4)(*([1 if "CrossAttn" in t else 0
                                                              for t in cfg["down_block_types"]] + [0] * (4 - n))),
                            layers_per_block=cfg["layers_per_block"], n_heads=cfg["attention_head_dim"],
                            cross_dim=cfg["cross_attention_dim"], text_len=text_len,
                            norm_groups=cfg["norm_num_groups"],
                            dtype=3 if self.precision == "bf16x3" else {torch.bfloat16: 0, torch.float16: 1, torch.float32: 2}[dtype],
                            latent_repeat=self.latent_repeat)
        h = C.c_void_p()
        _lib.check(_lib.lib().sdn_unet_create(C.byref(c), C.byref(h)), "sdn_unet_create")
        self._h = h
        self._weights = None
        self._ws = {}
        self.tail_split = False            # set_tail_split(): see _tail_split_of
        self._read_manifest()

    def _read_manifest(self):
        """diffusers-keyed tensors -> self.manifest; engine-derived regions (SDN_P_DERIVED) are left to _prepare()."""
        h = self._h
        self.manifest = []
        info = _lib.ParamInfo()
        for i in range(_lib.lib().sdn_unet_param_count(h)):
            _lib.check(_lib.lib().sdn_unet_param_info(h, i, C.byref(info)), "sdn_unet_param_info")
            if info.kind == P_DERIVED:
                continue
            self.manifest.append(dict(name=info.name.decode(), kind=info.kind, rows=info.rows, cols=info.cols,
                                      rows_padded=info.rows_padded, offset=info.offset))
        self.weight_bytes = _lib.lib().sdn_unet_weight_bytes(h)

    def _prepare(self):
        """Let the engine fill its derived weight regions (LayerNorm-folded projections) from the uploaded tensors."""
        _lib.check(_lib.lib().sdn_unet_prepare(self._h, _lib.dptr(self._weights), _lib.stream_ptr()), "sdn_unet_prepare")
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().sdn_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- parameters ---------------------------------------------------------------------------------
    def state_dict_shapes(self) -> dict:
        """diffusers key -> source tensor shape."""
        out = {}
        cin = self.config.in_channels
        for p in self.manifest:
            k, r, c = p["kind"], p["rows"], p["cols"]
            if k in (P_VEC_F32, P_GEGLU_VEC):
                out[p["name"]] = (r,)
            elif k == P_CONV3X3:
                out[p["name"]] = (r, c // 9, 3, 3)
            elif p["name"].endswith(("proj_in.weight", "proj_out.weight", "conv_shortcut.weight")):
                out[p["name"]] = (r, c, 1, 1)
            else:
                out[p["name"]] = (r, c)
        del cin
        return out

    @staticmethod
    def _is_norm_param(name: str) -> bool:
        return ".norm" in name or name.startswith("conv_norm_out")

    def synthetic_state_dict(self, seed: int = 1234) -> dict:
        """Random weights of this architecture (there are no checkpoints on the box): variance-preserving
        uniform U(-sqrt(3/fan_in), sqrt(3/fan_in)) for matrices, small uniform biases, norm gains near 1."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for name, shape in self.state_dict_shapes().items():
            if len(shape) == 1:
                if self._is_norm_param(name):
                    base = 1.0 if name.endswith("weight") else 0.0
                    sd[name] = base + 0.1 * (torch.rand(shape, generator=g) - 0.5)
                else:
                    sd[name] = 0.2 * (torch.rand(shape, generator=g) - 0.5)
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                bound = (3.0 / fan_in) ** 0.5
                sd[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        return sd

    def _pack_one(self, p: dict, t: torch.Tensor) -> torch.Tensor:
        """One state_dict tensor in the engine layout, as raw bytes (uint8, 1-D)."""
        t = t.detach().float().cpu()
        k = p["kind"]
        if k == P_CONV3X3:
            t = t.permute(0, 2, 3, 1).reshape(p["rows"], p["cols"])
        elif k in (P_MAT, P_GEGLU_MAT):
            t = t.reshape(p["rows"], p["cols"])
        if k in (P_GEGLU_MAT, P_GEGLU_VEC):
            t = _interleave16(t)
        if k not in (P_VEC_F32, P_GEGLU_VEC):
            t = t.to(self.dtype)
        return t.contiguous().view(torch.uint8).reshape(-1)

    def pack_state_dict(self, sd: dict) -> torch.Tensor:
        """CPU uint8 buffer in the engine layout."""
        buf = torch.zeros(self.weight_bytes, dtype=torch.uint8)
        for p in self.manifest:
            raw = self._pack_one(p, sd[p["name"]])
            buf[p["offset"]:p["offset"] + raw.numel()] = raw
        return buf

    def load_state_dict(self, sd: dict, device="cuda"):
        missing = [p["name"] for p in self.manifest if p["name"] not in sd]
        if missing:
            raise KeyError(f"state_dict lacks {len(missing)} keys, e.g. {missing[:3]}")
        _lib.require_gpu()
        if type(self).pack_state_dict is not UNet2DConditionModel.pack_state_dict:
            # a subclass with its own packing rules (MMDiT position-embedding crop, VAE 1x1 mixers, CLIP key prefixes)
            self._weights = self.pack_state_dict(sd).to(device)
            return self._prepare()
        # tensor by tensor into the device buffer: the engine-derived regions between them (LayerNorm-folded / product weights;
        # in the bf16x3 plan the expanded bf16 copy of every matrix, 1.5 x the f32 weights) never exist on the host
        buf = torch.zeros(self.weight_bytes, dtype=torch.uint8, device=device)
        for p in self.manifest:
            raw = self._pack_one(p, sd[p["name"]])
            buf[p["offset"]:p["offset"] + raw.numel()].copy_(raw)
        self._weights = buf
        return self._prepare()

    def load_synthetic_on_device(self, seed: int = 1234, device="cuda"):
        """Random weights generated DIRECTLY in the packed engine layout on the GPU (benchmarks: no checkpoints exist
        on the box and the 0.86 G-parameter CPU generate+pack path takes tens of seconds per rank).  Same distributions
        as synthetic_state_dict(); the values are not the CPU generator's, so parity tests use the state_dict path."""
        _lib.require_gpu()
        g = torch.Generator(device=device).manual_seed(seed)
        buf = torch.zeros(self.weight_bytes, dtype=torch.uint8, device=device)
        for p in self.manifest:
            n = p["rows"] * max(p["cols"], 1)
            if p["kind"] in (P_VEC_F32, P_GEGLU_VEC):
                is_gain = self._is_norm_param(p["name"]) and p["name"].endswith("weight")
                is_nb = self._is_norm_param(p["name"]) and p["name"].endswith("bias")
                amp = 0.1 if (is_gain or is_nb) else 0.2
                t = (torch.rand(n, generator=g, device=device) - 0.5) * amp + (1.0 if is_gain else 0.0)
                buf[p["offset"]:p["offset"] + 4 * n] = t.view(torch.uint8)
            else:
                t = ((torch.rand(n, generator=g, device=device) * 2 - 1) * (3.0 / max(p["cols"], 1)) ** 0.5).to(self.dtype)
                buf[p["offset"]:p["offset"] + t.element_size() * n] = t.view(torch.uint8)
        self._weights = buf
        return self._prepare()

    # ---- forward --------------------------------------------------------------------------------------
    def flops(self, batch: int):
        a = C.c_double()
        total = _lib.lib().sdn_unet_flops(self._h, batch, C.byref(a))
        return total, a.value

    def set_graph_mode(self, on: bool = True):
        """Replay each forward as one hipGraph (sdn_unet_set_graph_mode): worth it when the batch is small enough for the
        ~850 launches of a forward to be the bound (a single prompt: 13 ms -> a few ms per step).  Identical results."""
        _lib.lib().sdn_unet_set_graph_mode(self._h, 1 if on else 0)
        return self

    def set_text_version(self, version: int = 0):
        """Declare the contents of the text operand of the next forwards (sdn_unet_set_text_version): while the version, the
        batch and the buffers stay the same, the cross-attention K / V projections of the text are not recomputed.  0 = undeclared."""
        _lib.lib().sdn_unet_set_text_version(self._h, int(version))
        self._text_version = int(version)
        return self

    def set_split_k(self, on: bool = True):
        """Small-batch option (sdn_unet_set_split_k): under-filled GEMMs run in split-K form.  Rebuilds the plans, so the
        cached workspaces are dropped.  Off by default (keeps outputs bit-identical across batch sizes)."""
        _lib.lib().sdn_unet_set_split_k(self._h, 1 if on else 0)
        self._ws = {}
        return self

    def profile_next(self):
        """Arm HIP-event profiling of the next forward (diagnostics; see sdn_unet_profile_next)."""
        _lib.lib().sdn_unet_profile_next(self._h)

    def profile_read(self) -> list:
        rows = (_lib.ProfileRow * 32)()
        n = _lib.lib().sdn_unet_profile_read(self._h, rows, 32)
        if n < 0:
            raise _lib.SdnError("sdn_unet_profile_read failed (no profiled forward?)")
        return [dict(kernel=rows[i].kernel.decode(), launches=rows[i].launches, ms=rows[i].ms, flops=rows[i].flops,
                     bytes=rows[i].bytes) for i in range(n)]

    def _workspace(self, batch: int, device):
        ws = self._ws.get(batch)
        if ws is None:
            n = _lib.lib().sdn_unet_workspace_bytes(self._h, batch)
            ws = torch.empty(n, dtype=torch.uint8, device=device)
            self._ws[batch] = ws
        return ws

    def prepare_text(self, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        e = encoder_hidden_states
        if e.shape[1] != self.text_len or e.shape[2] != self.config.cross_attention_dim:
            raise _lib.SdnError(f"encoder_hidden_states must be [B,{self.text_len},{self.config.cross_attention_dim}]")
        return e.to(self.dtype).contiguous()

    # ---- a few latents over whole waves of tiles: the tail runs beside the main forward ------------------------------------
    # Every level of the SD-v1.4 plan is a whole number of 256-CU waves of 256-row tiles when the batch is a multiple of 64
    # samples (64^2 level: 64 x 4096 / 256 = 1024 tiles = 4 waves; 16^2: 64 x 4 = 256 tiles = 1 wave).  A batch a few latents over
    # such a multiple (8 ranks x 515 prompts: 65 prompts x 3 branches = 195 samples on three of the ranks) starts one more wave in
    # every GEMM / conv launch for a handful of tiles: 162.6 ms against 152.7 ms for 192 samples, where the work grew 1.6 %
    # (tools/ab_tail_split.py).  With `tail_split` on, such a forward runs as the aligned part on the caller's stream and the
    # remaining latents as their own small forward (a second handle over the SAME packed weights) on a side stream, joined before
    # the outputs are put back in branch-major order: 156.8 ms, bit-identical outputs on the 16-bit plans (rows are independent;
    # bf16x3: 2e-6, its row chunks differ).  DESIGN 10.13.
    TAIL_MAX_SAMPLES = 12            # measured: 3 samples -3.6 %, 24 samples -1.4 % of the unsplit forward; above this it stays one batch

    def set_tail_split(self, on: bool = True):
        self.tail_split = bool(on)
        return self

    def _tail_split_of(self, b: int):
        """(latents in the aligned main forward, latents in the tail), or None when the batch runs as one forward."""
        rep = self.latent_repeat
        if not getattr(self, "tail_split", False) or rep <= 1 or type(self) is not UNet2DConditionModel:
            return None
        p = b // rep
        q = 64 // math.gcd(64, rep)                                  # latents per 64-sample quantum
        pm = p // q * q
        r = p - pm
        if pm == 0 or r == 0 or r * rep > self.TAIL_MAX_SAMPLES:
            return None
        return pm, r

    def max_samples(self) -> int:
        """Largest batch ONE launch plan addresses: the LDS-DMA tiles carry 31-bit byte offsets per operand, and the widest 16-bit
        operand of a forward is an up-path resnet's input (skip connection concatenated: 64^2 x 960 channels = 7.9 MB per sample
        for SD-v1.4 -> 273 samples).  Larger batches run as several forwards (`_chunks_of`)."""
        c = self.config
        boc = list(c.block_out_channels)
        n = len(boc)
        worst = max((c.sample_size >> i) ** 2 * (boc[min(i + 1, n - 1)] + boc[i]) * 2 for i in range(n))
        cap = ((1 << 31) - 1) // worst
        if self.precision in ("fp32", "bf16x3"):
            # fp32-storage plans: their GEMMs cut launches themselves (6-byte triples), the other kernels index 4-byte tensors of twice
            # the size; exercised up to 195 samples at full size (tests, tools/ab_tail_split.py) -- stay where that evidence is
            cap = min(cap, 3 * (cap // 4))
        return cap

    def _chunks_of(self, b: int):
        """How a forward of b samples runs: None = one launch plan; else [(first latent, latents, on the side stream)].  Two reasons to
        cut: the tail rule above, and a batch above `max_samples()` (it used to fail with SDN_E_INVALID): whole-wave chunks (multiples of
        64 samples) one after the other on the caller's stream, each on its own handle (its own workspace and text K / V cache), a
        small remainder beside them."""
        if type(self) is not UNet2DConditionModel:
            return None
        rep = self.latent_repeat
        p = b // rep
        cap = self.max_samples() // rep                                # latents one plan takes
        if p <= cap:
            ts = self._tail_split_of(b)
            return None if ts is None else [(0, ts[0], False), (ts[0], ts[1], True)]
        q = 64 // math.gcd(64, rep)
        per = cap // q * q if cap >= q else cap                        # whole waves of tiles where the cap allows it
        chunks = [(lo, min(per, p - lo), False) for lo in range(0, p, per)]
        lo, n, _ = chunks[-1]
        if n * rep <= self.TAIL_MAX_SAMPLES and getattr(self, "tail_split", False) and rep > 1:
            chunks[-1] = (lo, n, True)
        return chunks

    def _forward_chunks(self, sample, timestep, text, out, chunks):
        rep, dev = self.latent_repeat, sample.device
        p = sample.shape[0]
        st = getattr(self, "_split", None)
        if st is None or st["key"] != (p, tuple(chunks), dev):
            kw = dict(text_len=self.text_len, dtype=self.dtype, latent_repeat=rep,
                      precision=self.precision if self.precision in ("fp32", "bf16x3") else None, **vars(self.config))
            tshape, oshape = tuple(text.shape[1:]), tuple(out.shape[1:])
            parts = [dict(lo=lo, n=n, side=side, net=self if i == 0 else UNet2DConditionModel(**kw),
                          # rep = 1: a chunk's text / output rows are contiguous views of the caller's tensors, nothing is staged
                          text=None if rep == 1 else torch.empty((rep * n,) + tshape, dtype=self.dtype, device=dev),
                          out=None if rep == 1 else torch.empty((rep * n,) + oshape, dtype=torch.float32, device=dev))
                     for i, (lo, n, side) in enumerate(chunks)]
            st = self._split = dict(key=(p, tuple(chunks), dev), parts=parts, text_key=None,
                                    side=torch.cuda.Stream(device=dev) if any(c[2] for c in chunks) else None)
        ver = int(getattr(self, "_text_version", 0))
        tkey = (ver, text.data_ptr())
        if rep > 1 and (ver == 0 or st["text_key"] != tkey):         # branch-major text rows [rep][p] -> [rep][n] per chunk
            tv = text.view(rep, p, *text.shape[1:])
            for c in st["parts"]:
                c["text"].view(rep, c["n"], *text.shape[1:]).copy_(tv[:, c["lo"]:c["lo"] + c["n"]])
            st["text_key"] = tkey
        cur = torch.cuda.current_stream(dev)
        if st["side"] is not None:
            st["side"].wait_stream(cur)                              # latents and text are ready where the caller's stream stands

        def run(c):
            net, lo, n = c["net"], c["lo"], c["n"]
            if net is not self:
                net._weights = self._weights                         # the derived regions live in the buffer: nothing to prepare again
                net.set_text_version(ver)
            if rep == 1:
                net._forward_one(sample[lo:lo + n], timestep, text[lo:lo + n], out[lo:lo + n])
            else:
                net._forward_one(sample[lo:lo + n], timestep, c["text"], c["out"])

        for c in st["parts"]:
            if c["side"]:
                with torch.cuda.stream(st["side"]):
                    run(c)
        for c in st["parts"]:
            if not c["side"]:
                run(c)
        if st["side"] is not None:
            cur.wait_stream(st["side"])
        if rep > 1:
            ov = out.view(rep, p, -1)
            for c in st["parts"]:
                ov[:, c["lo"]:c["lo"] + c["n"]].copy_(c["out"].view(rep, c["n"], -1))
        return out

    def forward_into(self, sample, timestep, text_bf16, out):
        """No-allocation form used by the engine loop (text already bf16, `out` preallocated fp32).  With latent_repeat = r
        the sample has B / r rows, text and out have B."""
        b = text_bf16.shape[0]
        if sample.shape[0] * self.latent_repeat != b or out.shape[0] != b:
            raise _lib.SdnError(f"batch mismatch: {sample.shape[0]} latents x latent_repeat {self.latent_repeat} vs {b} text rows")
        chunks = self._chunks_of(b)
        if chunks is not None:
            _lib.dptr(sample, torch.float32), _lib.dptr(text_bf16, self.dtype), _lib.dptr(out, torch.float32)     # the loud checks
            return self._forward_chunks(sample, timestep, text_bf16, out, chunks)
        return self._forward_one(sample, timestep, text_bf16, out)

    def _forward_one(self, sample, timestep, text_bf16, out):
        b = text_bf16.shape[0]
        if type(self) is UNet2DConditionModel and b > self.max_samples():
            raise _lib.SdnError(f"{b} samples exceed what one launch plan addresses ({self.max_samples()}): forward_into cuts such batches")
        ws = self._workspace(b, sample.device)
        _lib.check(_lib.lib().sdn_unet_forward(self._h, _lib.dptr(self._weights), _lib.dptr(sample, torch.float32),
                                               float(timestep), _lib.dptr(text_bf16, self.dtype),
                                               _lib.dptr(out, torch.float32), b, _lib.dptr(ws), ws.numel(),
                                               _lib.stream_ptr()), "sdn_unet_forward")
        return out

    def __call__(self, sample, timestep, encoder_hidden_states=None, return_dict=True, **unused):
        _lib.require_gpu()
        if self._weights is None:
            raise _lib.SdnError("no weights loaded: call load_state_dict() first")
        x = sample.float().contiguous()
        s = self.config.sample_size
        if tuple(x.shape[1:]) != (self.config.in_channels, s, s):
            raise _lib.SdnError(f"sample must be [B,{self.config.in_channels},{s},{s}], got {tuple(x.shape)}")
        e = self.prepare_text(encoder_hidden_states)
        if e.shape[0] != x.shape[0] * self.latent_repeat:
            raise _lib.SdnError("batch mismatch between sample and encoder_hidden_states")
        out = torch.empty((e.shape[0], self.config.out_channels, s, s), dtype=torch.float32, device=x.device)
        self.forward_into(x, float(timestep), e, out)
        return UNetOutput(out) if return_dict else (out,)
