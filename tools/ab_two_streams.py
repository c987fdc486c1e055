"""Does the forward gain from running the batch as two half-batches on two HIP streams (kernels of one stream filling the other's
tails / prologues / epilogues)?  One process, same weights buffer, two plans' worth of workspace.
    python tools/ab_two_streams.py [B=192] [mode=bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_denoiser_amd.unet import UNet2DConditionModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 192
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
kw = dict(precision="bf16x3") if mode == "bf16x3" else dict(dtype=torch.bfloat16)
rep = 3


def mk():
    u = UNet2DConditionModel(latent_repeat=rep, **kw)
    u.load_synthetic_on_device(1234)
    return u


def time_it(fn, n=6):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


u = mk()
x = torch.randn(B // rep, 4, 64, 64, device="cuda")
tb = u.prepare_text(torch.randn(B, 77, 768, device="cuda"))
y = torch.empty(B, 4, 64, 64, device="cuda")
one = time_it(lambda: u.forward_into(x, 981.0, tb, y))
print(f"{mode}: one stream, B = {B}: {one:.2f} ms per forward", flush=True)

# two half batches on two streams (second handle: its own workspace; the same packed weights)
u2 = mk()
h = B // 2
xs = [x[:h // rep].contiguous(), x[h // rep:].contiguous()]
# branch-major text rows: [uncond | text' | text] blocks of B/3 -> per half: rows of each branch
tbv = tb.view(rep, B // rep, *tb.shape[1:])
tbs = [tbv[:, :h // rep].reshape(h, *tb.shape[1:]).contiguous(), tbv[:, h // rep:].reshape(h, *tb.shape[1:]).contiguous()]
ys = [torch.empty(h, 4, 64, 64, device="cuda") for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
us = [u, u2]


def two():
    cur = torch.cuda.current_stream()
    for i, st in enumerate((s1, s2)):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            us[i].forward_into(xs[i], 981.0, tbs[i], ys[i])
    for st in (s1, s2):
        cur.wait_stream(st)


half = time_it(lambda: u.forward_into(xs[0], 981.0, tbs[0], ys[0]))
both = time_it(two)
print(f"{mode}: one stream, B = {h}: {half:.2f} ms; two streams x B = {h}: {both:.2f} ms per pair  (vs {one:.2f} for the single B = {B} forward: {100 * (one / both - 1):+.1f} %)")
