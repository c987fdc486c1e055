"""Front-end with the semantics of the reference's repellency/repellency_methods_fast_sdv3.py
(run_nudity_sdv3.py:32, run_coco30k_sdv3.py:39): the fast module plus per-pixel channel normalisation of the
QUERY before the distance (fast_sdv3:150-152,192-194,238-240,331-333); the update is applied to the
un-normalised x.  A zero-norm pixel makes the whole output NaN, as in the reference."""
from __future__ import annotations

from . import repellency_methods_fast as _fast
from ._engine import QNORM_CHANNEL, make_registry

__CONDITIONING_METHOD__, register_conditioning_method, get_repellency_method = make_registry()


class RepellencyMethod(_fast.RepellencyMethod):
    qnorm = QNORM_CHANNEL


@register_conditioning_method(name="kernel_fast")
class RBFKernelRepellency(_fast.RBFKernelRepellency):
    qnorm = QNORM_CHANNEL


@register_conditioning_method(name="sparse")
class SparseRepellency(_fast.SparseRepellency):
    qnorm = QNORM_CHANNEL


@register_conditioning_method(name="random_noise")
class RandomNoiseRepellency(_fast.RandomNoiseRepellency):
    qnorm = QNORM_CHANNEL
