#!/usr/bin/env python3
"""SD3-medium MMDiT forward timing (fp16 storage, as the reference runs SD-v3): SIDE=64 (512x512) or 128 (1024x1024)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.mmdit import SD3Transformer2DModel  # noqa: E402

SIDE, B = int(os.environ.get("SIDE", "128")), int(os.environ.get("B", "8"))
m = SD3Transformer2DModel(sample_size=SIDE)
m.load_synthetic_on_device(3)
x = torch.randn(B, 16, SIDE, SIDE, device="cuda"); e = torch.randn(B, 333, 4096, device="cuda"); pl = torch.randn(B, 2048, device="cuda")
m(x, timestep=900.0, encoder_hidden_states=e, pooled_projections=pl)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); m(x, timestep=900.0, encoder_hidden_states=e, pooled_projections=pl); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts)[2]
fl, at = m.flops(B)
print(f"SD3-medium side {SIDE} B={B}: {t:.2f} ms per forward ({t / B:.2f} ms per sample), {fl / t / 1e9:.1f} TFLOP/s, attention share {at / fl:.2f}")
m.profile_next(); m(x, timestep=900.0, encoder_hidden_states=e, pooled_projections=pl)
rows = sorted(m.profile_read(), key=lambda r: -r["ms"]); tot = sum(r["ms"] for r in rows)
for r in rows[:12]:
    print(f"  {r['kernel']:18s} x{r['launches']:4d} {r['ms']:8.3f} ms {100 * r['ms'] / tot:5.1f} %  {(r['flops'] / r['ms'] / 1e9) if r['flops'] else 0:7.1f} TF/s")
if os.environ.get("SHAPES"):                     # per (kernel, M, N, K): HIP events around every launch of three forwards
    import collections
    import ctypes as C
    import safe_denoiser_amd as sda
    lib = sda.lib()
    lib.sdn_debug_profile_ops.restype = C.c_int
    lib.sdn_debug_profile_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    agg = collections.OrderedDict()
    for _ in range(3):
        m.profile_next(); m(x, timestep=900.0, encoder_hidden_states=e, pooled_projections=pl)
        out = (C.c_double * (6 * 32768))(); lab = C.create_string_buffer(24 * 32768)
        n = lib.sdn_debug_profile_ops(m._h, out, lab, 32768)
        for i in range(n):
            name = lab.raw[i * 24:(i + 1) * 24].split(b"\0")[0].decode()
            a = agg.setdefault((name, int(out[i * 6 + 3]), int(out[i * 6 + 4]), int(out[i * 6 + 5])), [0, 0.0, 0.0])
            a[0] += 1; a[1] += out[i * 6]; a[2] += out[i * 6 + 1]
    tot = sum(a[1] for a in agg.values()) / 3
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "24"))]:
        ms = a[1] / 3
        print(f"  {key[0]:16s} M={key[1]:7d} N={key[2]:6d} K={key[3]:6d}  x{a[0] // 3:3d}  {ms:7.3f} ms  {100 * ms / tot:5.1f}%  "
              f"{(a[2] / 3 / (ms * 1e-3) / 1e12) if ms > 0 else 0:7.1f} TF/s")
