#!/usr/bin/env python3
"""Per-launch profile of one UNet forward (HIP events recorded by the plan runner), grouped by (kernel, shape)."""
import collections
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import safe_denoiser_amd as sda  # noqa: E402
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(os.environ.get("B", "64"))
u = UNet2DConditionModel()
buf = torch.zeros(u.weight_bytes, dtype=torch.uint8, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
for p in u.manifest:
    n = p["rows_padded"] * max(p["cols"], 1)
    if p["kind"] in (0, 4):
        t = (torch.rand(n, generator=g, device="cuda") - 0.5) * 0.2 + (1.0 if "norm" in p["name"] and p["name"].endswith("weight") else 0.0)
        buf[p["offset"]:p["offset"] + 4 * n] = t.view(torch.uint8)
    else:
        t = ((torch.rand(n, generator=g, device="cuda") * 2 - 1) * (3.0 / max(p["cols"], 1)) ** 0.5).bfloat16()
        buf[p["offset"]:p["offset"] + 2 * n] = t.view(torch.uint8)
u._weights = buf
u._prepare()
if os.environ.get("LN_FOLD") == "0":
    sda.lib().sdn_debug_set_ln_fold(u._h, 0)
if os.environ.get("GN_FUSE") == "0":
    sda.lib().sdn_debug_set_gn_fuse(u._h, 0)
if os.environ.get("SPLIT_K"):
    u.set_split_k(True)
if os.environ.get("SUBBATCH") is not None:
    sda.lib().sdn_debug_set_subbatch_bytes(u._h, C.c_longlong(int(os.environ["SUBBATCH"])))
x = torch.randn(B, 4, 64, 64, device="cuda")
e = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty_like(x)
u.forward_into(x, 981.0, e, y)
agg = collections.OrderedDict()
lib = sda.lib()
lib.sdn_debug_profile_ops.restype = C.c_int
lib.sdn_debug_profile_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
for _ in range(3):
    u.profile_next()
    u.forward_into(x, 981.0, e, y)
    out = (C.c_double * (6 * 32768))()
    lab = C.create_string_buffer(24 * 32768)
    n = lib.sdn_debug_profile_ops(u._h, out, lab, 32768)
    for i in range(n):
        name = lab.raw[i * 24:(i + 1) * 24].split(b"\0")[0].decode()
        key = (name, int(out[i * 6 + 3]), int(out[i * 6 + 4]), int(out[i * 6 + 5]))
        a = agg.setdefault(key, [0, 0.0, 0.0])
        a[0] += 1; a[1] += out[i * 6]; a[2] += out[i * 6 + 1]
tot = sum(a[1] for a in agg.values()) / 3
print(f"B={B}: {tot:.2f} ms per forward")
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", "28"))]:
    ms = a[1] / 3
    tf = a[2] / 3 / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{key[0]:16s} M={key[1]:7d} N={key[2]:6d} K={key[3]:6d}  x{a[0] // 3:3d}  {ms:7.3f} ms  {100 * ms / tot:5.1f}%  {tf:7.1f} TF/s")
