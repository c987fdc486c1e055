#!/usr/bin/env python3
"""A batch that is a few prompts over whole waves of 256-row tiles (8 ranks x 515 prompts: 65 prompts = 195 samples on three
ranks, against the 192 that fill every level's grid exactly): one forward at B, against the B_main forward on the launch stream
with the remaining prompts' forward (its own handle, the same packed weights) on a second stream -- the product's path,
`UNet2DConditionModel.set_tail_split` (on by default in SafeDenoiserPipeline).  End to end: `python bench.py --prompts-per-batch 65
[--no-tail-split]`.
    python tools/ab_tail_split.py [prompts=65] [main=64] [mode=bf16|f16|bf16x3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 65
PM = int(sys.argv[2]) if len(sys.argv) > 2 else 64
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
kw = dict(precision="bf16x3") if mode == "bf16x3" else dict(dtype=torch.float16 if mode == "f16" else torch.bfloat16)
rep = 3
N = int(os.environ.get("N", "8"))


def time_it(fn, n=N):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


u = UNet2DConditionModel(latent_repeat=rep, **kw)
u.load_synthetic_on_device(1234)
x = torch.randn(P, 4, 64, 64, device="cuda")
tb = u.prepare_text(torch.randn(rep * P, 77, 768, device="cuda"))
y = torch.empty(rep * P, 4, 64, 64, device="cuda")
u.set_text_version(1)
one = time_it(lambda: u.forward_into(x, 981.0, tb, y))
print(f"{mode}: one forward, {P} prompts (B = {rep * P}): {one:.2f} ms", flush=True)

tbm = tb.view(rep, P, *tb.shape[1:])[:, :PM].reshape(rep * PM, *tb.shape[1:]).contiguous()
ym = torch.empty(rep * PM, 4, 64, 64, device="cuda")
u.set_text_version(2)
main = time_it(lambda: u.forward_into(x[:PM], 981.0, tbm, ym))
print(f"{mode}: one forward, {PM} prompts: {main:.2f} ms", flush=True)

# the product path: UNet2DConditionModel.set_tail_split (the aligned part on this stream, the tail on a side stream)
u.set_tail_split(True)
assert u._tail_split_of(rep * P) == (PM, P - PM), u._tail_split_of(rep * P)
y2 = torch.empty_like(y)
u.set_text_version(3)
both = time_it(lambda: u.forward_into(x, 981.0, tb, y2))
err = float((y2 - y).norm() / y.norm())
print(f"{mode}: aligned part + tail of {P - PM} as two concurrent forwards (set_tail_split): {both:.2f} ms  ({100 * (both / one - 1):+.1f} % vs the "
      f"single forward; if the extra prompts cost their share: {100 * (main * P / PM / one - 1):+.1f} %); outputs vs the single forward: {err:.2e}",
      flush=True)
