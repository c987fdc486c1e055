"""Per-prompt random streams, drawn for a whole batch in one launch.

Every prompt of a batch keeps its OWN torch.Generator (seeded like the reference's `gen.manual_seed(seed)`, run_nudity.py:448),
so that the sequence it sees -- latents, the x0 probe's discarded variance draw, the conditional re-noise draw, the step's
variance draw -- is exactly what a one-prompt call of the reference pipeline would draw.  `BatchedNormal.draw` produces what
`torch.randn(shape, generator=g_p, device=dev)` would return for each selected prompt with ONE sdn_randn_philox launch (Philox
seed / offset read from the generator, offsets advanced with Generator.set_offset), instead of P `torch.randn` launches + P
slice copies.  The kernel is checked against torch itself the first time a shape is used on a device (`self_check`); if the
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

    def draw(self, generators: Sequence[torch.Generator], out: torch.Tensor, which: Optional[Sequence[int]] = None,
             shape=None):
        """out[p] <- randn(shape) of generators[p] for p in `which` (default: all); other rows are left untouched.
        `out` is [P, ...] fp32 contiguous with prod(shape[1:]) == numel."""
        idx = list(range(len(generators))) if which is None else list(which)
        if not idx:
            return out
        if not self.ok:
            shp = shape or (1,) + tuple(out.shape[1:])
            for p in idx:
                out[p:p + 1] = torch.randn(shp, generator=generators[p], device=self.device, dtype=torch.float32)
            return out
        seeds = [self._signed(generators[p].initial_seed()) for p in idx]
        offs = [generators[p].get_offset() for p in idx]
        self._launch(seeds, offs, None if which is None else idx, out)
        for p, o in zip(idx, offs):
            generators[p].set_offset(o + self.increment)
        return out

    def skip(self, generators: Sequence[torch.Generator], which: Optional[Sequence[int]] = None):
        """A draw whose values nobody reads (the x0 probe's scheduler.step variance noise): advance the streams only."""
        idx = range(len(generators)) if which is None else which
        if not self.ok:
            for p in idx:
                torch.randn(self.numel, generator=generators[p], device=self.device, dtype=torch.float32)
            return
        for p in idx:
            generators[p].set_offset(generators[p].get_offset() + self.increment)
