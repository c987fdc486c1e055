"""safe_denoiser_amd -- MI355X-native engine for the safe-denoiser sampling hot path.

Host orchestration is Python; every arithmetic step of the hot path runs in hand-written HIP kernels
(gfx950) inside ``libsdn.so`` behind the C ABI declared in ``include/sdn.h``.  There is NO CPU fallback:
importing works anywhere (so CPU-only tooling can inspect the package), but any compute call raises
``SdnUnavailable`` when the library or a gfx950 device is missing.
"""
from ._lib import SdnUnavailable, SdnError, lib, lib_path  # noqa: F401

__all__ = ["SdnUnavailable", "SdnError", "lib", "lib_path"]
