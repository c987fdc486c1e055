"""CPU oracle for the safe-denoiser hot path.

TEST INFRASTRUCTURE ONLY. Nothing in ``safe_denoiser_amd`` (the product) may
import from here; the only legitimate importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``, and
there only as the checker / the timed CPU baseline.

Parity pinning status (see DESIGN.md section "Oracle"):
  * repellency (rows R1-R6)  -- PINNED by golden vectors captured from the
    reference's own ``repellency.*`` modules (tests/golden/make_golden.py).
  * SAFREE text projection (8f row 1, SD-v1.4 and SD-v3 variants) -- PINNED by golden vectors produced by
    executing the reference's own helper functions (tests/golden/make_safree_golden.py).
  * CLIP text encoder -- PINNED against transformers.CLIPTextModel (tests/golden/make_clip_golden.py).
  * schedulers / UNet / loop (rows S1-S3, U1-U6, P1-P3) -- PARITY UNPINNED at
    the reference level: the arithmetic lives in diffusers==0.29.0, which is
    absent from /root/reference and from this image, and the reference ships
    no tests or golden vectors for it.  The restatement follows the published
    diffusers-0.29.0 definitions and the wiring spec vendored in the reference
    (file:line cited per function) and is checked by self-consistency KATs.
"""
