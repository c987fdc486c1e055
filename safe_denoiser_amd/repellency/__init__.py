"""Repellency plug-in front-ends (same module names, registry and call surface as the reference's
``repellency`` package); all arithmetic runs in libsdn (sdn_repel_apply / sdn_repel_calibrate)."""
from . import repellency_methods_fast, repellency_methods_fast_sdv3, repellency_methods_threshold  # noqa: F401
