"""Per-prompt random streams, drawn for a whole batch in one launch.

Every prompt of a batch keeps its OWN torch.Generator (seeded like the reference's `gen.manual_seed(seed)`, run_nudity.py:448),
so that the sequence it sees -- latents, the x0 probe's discarded variance draw, the conditional re-noise draw, the step's
variance draw -- is exactly what a one-prompt call of the reference pipeline would draw.  `BatchedNormal.draw` produces what
`torch.randn(shape, generator=g_p, device=dev)` would return for each selected prompt with ONE launch instead of P `torch.randn`
launches + P slice copies.  `bind(generators)` uploads the (seed, philox offset) pairs ONCE per pipeline call; after that the
offsets live and advance on the device (sdn_randn_philox_state) and the conditional re-noise draw is selected by the loop's
device-side is_negation vector, so a draw moves nothing between host and device -- the torch.Generator objects are kept in step
with `set_offset` on the host (pure host bookkeeping; a generator that was touched by someone else in between is detected by
its offset and re-uploaded).  The kernel is checked against torch itself the first time a shape is used on a device (`self_check`); if the
bits ever differed (another torch build, another hipRAND) the class falls back to the per-prompt torch path and says so.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Optional, Sequence

import torch

from . import _lib


class BatchedNormal:
    def __init__(self, device, numel: int):
        self.device, self.numel = device, int(numel)
        g, inc = C.c_int32(), C.c_int64()
        _lib.check(_lib.lib().sdn_randn_philox_plan(self.numel, C.byref(g), C.byref(inc)), "sdn_randn_philox_plan")
        self.increment = int(inc.value)
        self._host_stale = False
        self.ok = self._self_check()

    def _launch(self, seeds, offsets, rows, out):
        meta = torch.tensor([seeds, offsets], dtype=torch.int64).to(self.device, non_blocking=False)
        r = None if rows is None else torch.tensor(rows, dtype=torch.int32).to(self.device)
        _lib.check(_lib.lib().sdn_randn_philox(meta[0].data_ptr(), meta[1].data_ptr(), None if r is None else r.data_ptr(),
                                               len(seeds), self.numel, out.data_ptr(), _lib.stream_ptr()), "sdn_randn_philox")

    def _self_check(self) -> bool:
        """One draw from a scratch generator, kernel vs torch.randn, then the generator's own bookkeeping (offset advance)."""
        try:
            g = torch.Generator(device=self.device).manual_seed(987654321)
            torch.randn(7, generator=g, device=self.device)                       # a non-zero starting offset
            seed, off = g.initial_seed(), g.get_offset()
            ref = torch.randn(self.numel, generator=g, device=self.device)
            out = torch.empty(1, self.numel, dtype=torch.float32, device=self.device)
            self._launch([self._signed(seed)], [off], None, out)
            same = bool(torch.equal(out[0], ref)) and g.get_offset() == off + self.increment
        except Exception as e:                                                    # pragma: no cover (defensive)
            warnings.warn(f"sdn_randn_philox self-check raised {e!r}; using per-prompt torch.randn")
            return False
        if not same:
            warnings.warn("sdn_randn_philox does not reproduce torch.randn on this build; using per-prompt torch.randn")
        return same

    @staticmethod
    def _signed(v: int) -> int:
        return v - 2 ** 64 if v >= 2 ** 63 else v

    # ---- device-resident generator states -------------------------------------------------------------------------
    def bind(self, generators: Sequence[torch.Generator]):
        """Upload (seed, offset) of every generator once; later draws run from (and advance) the device copy."""
        self.sync_host()                                                          # a previous binding's pending advances first
        self._gens = list(generators)
        self._mirror = [g.get_offset() for g in self._gens]
        meta = torch.tensor([[self._signed(g.initial_seed()) for g in self._gens], self._mirror], dtype=torch.int64)
        self._state = meta.to(self.device)                                        # [2, P]: seeds | offsets
        return self

    def _bound(self, generators) -> bool:
        gs = getattr(self, "_gens", None)
        if gs is None or len(gs) != len(generators) or any(a is not b for a, b in zip(gs, generators)):
            return False
        if self._host_stale:               # inside a sync-free loop: the device copy IS the state, the host objects are behind
            return True
        return all(g.get_offset() == m for g, m in zip(gs, self._mirror))       # nobody else drew from them in between

    def draw_flagged(self, generators: Sequence[torch.Generator], out: torch.Tensor, flags_dev: torch.Tensor):
        """out[p] <- randn of generators[p] for the rows whose DEVICE-side flag is non-zero, without the host knowing which:
        rows draw and advance on the device only; the torch.Generator objects are behind until `sync_host()`."""
        if not self.ok:
            raise _lib.SdnError("draw_flagged needs the batched kernel (self-check failed: use draw() with host flags)")
        if not self._bound(generators):
            self.bind(generators)
        _lib.check(_lib.lib().sdn_randn_philox_state(self._state[0].data_ptr(), self._state[1].data_ptr(), flags_dev.data_ptr(),
                                                     len(generators), self.numel, out.data_ptr(), _lib.stream_ptr()),
                   "sdn_randn_philox_state")
        self._host_stale = True
        return out

    def sync_host(self):
        """Bring the torch.Generator objects (and the host mirror) to the offsets the device-side streams have reached: ONE
        readback of P offsets, after a loop of draw_flagged calls."""
        if not getattr(self, "_host_stale", False) or getattr(self, "_gens", None) is None:
            self._host_stale = False
            return
        offs = self._state[1].cpu().tolist()
        for p, (g, o) in enumerate(zip(self._gens, offs)):
            self._mirror[p] = int(o)
            g.set_offset(int(o))
        self._host_stale = False

    def _advance_host(self, idx):
        if self._host_stale:               # the host is behind anyway; sync_host() reads the device's offsets
            return
        for p in idx:
            self._mirror[p] += self.increment
            self._gens[p].set_offset(self._mirror[p])

    def draw(self, generators: Sequence[torch.Generator], out: torch.Tensor, which: Optional[Sequence[int]] = None,
             shape=None, flags_dev: Optional[torch.Tensor] = None):
        """out[p] <- randn(shape) of generators[p] for p in `which` (default: all); other rows are left untouched.
        `out` is [P, ...] fp32 contiguous with prod(shape[1:]) == numel.  `flags_dev`: int32[P] on the device whose non-zero
        entries are exactly `which` (the loop's is_negation vector) -- then no index list is uploaded."""
        idx = list(range(len(generators))) if which is None else list(which)
        if not idx:
            return out
        if not self.ok:
            shp = shape or (1,) + tuple(out.shape[1:])
            for p in idx:
                out[p:p + 1] = torch.randn(shp, generator=generators[p], device=self.device, dtype=torch.float32)
            return out
        if which is None or flags_dev is not None:
            if not self._bound(generators):
                self.bind(generators)
            f = None if which is None else flags_dev
            _lib.check(_lib.lib().sdn_randn_philox_state(self._state[0].data_ptr(), self._state[1].data_ptr(),
                                                         None if f is None else f.data_ptr(), len(generators), self.numel,
                                                         out.data_ptr(), _lib.stream_ptr()), "sdn_randn_philox_state")
            self._advance_host(idx)
            return out
        # an arbitrary subset without a device-side flag vector: the index-list form (uploads seeds / offsets / rows)
        self.sync_host()
        seeds = [self._signed(generators[p].initial_seed()) for p in idx]
        offs = [generators[p].get_offset() for p in idx]
        self._launch(seeds, offs, idx, out)
        for p, o in zip(idx, offs):
            generators[p].set_offset(o + self.increment)
        self._gens = None                                                         # the device copy is stale now
        return out

    def draw_sequence(self, generator: torch.Generator, out: torch.Tensor, shape=None):
        """out[p] <- the p-th of out.shape[0] CONSECUTIVE `torch.randn(shape, generator=generator)` draws (one generator, the
        draws a Python loop over prompts would make one after the other -- the SD-v3 loop's `randn_like` on the global
        generator, models/sdv3/safe_denoiser_pipeline.py:1159) in one launch: draw p starts at the generator's offset + p x the
        per-draw increment."""
        n = out.shape[0]
        if not self.ok:
            shp = shape or (1,) + tuple(out.shape[1:])
            for p in range(n):
                out[p:p + 1] = torch.randn(shp, generator=generator, device=self.device, dtype=torch.float32)
            return out
        self.sync_host()
        off0 = generator.get_offset()
        seed = self._signed(generator.initial_seed())
        self._launch([seed] * n, [off0 + p * self.increment for p in range(n)], None, out)
        generator.set_offset(off0 + n * self.increment)
        self._gens = None
        return out

    def skip(self, generators: Sequence[torch.Generator], which: Optional[Sequence[int]] = None):
        """A draw whose values nobody reads (the x0 probe's scheduler.step variance noise): advance the streams only."""
        idx = range(len(generators)) if which is None else which
        if not self.ok:
            for p in idx:
                torch.randn(self.numel, generator=generators[p], device=self.device, dtype=torch.float32)
            return
        if which is None and self._bound(generators):
            _lib.check(_lib.lib().sdn_randn_philox_state(self._state[0].data_ptr(), self._state[1].data_ptr(), None,
                                                         len(generators), self.numel, None, _lib.stream_ptr()),
                       "sdn_randn_philox_state")
            self._advance_host(range(len(generators)))
            return
        self.sync_host()
        for p in idx:
            generators[p].set_offset(generators[p].get_offset() + self.increment)
        self._gens = None
