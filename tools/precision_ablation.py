"""Rounding-point ablation of the SD-v1.4 UNet (VERDICT round 2, next #1a): which classes of 16-bit rounding points carry the
distance of the 16-bit engine modes from the fp32 reference (run_nudity.py:277 runs fp32)?

Not a pytest file (no test_ prefix): run on the GPU box as `python tools/precision_ablation.py` -- the ORACLE (plain torch
ops, oracle/unet.py) is evaluated on the GPU through torch for speed; no libsdn kernel is involved.  For every arm the
full-size SD-v1.4 oracle (synthetic weights, seed 1234) is run with the rounding classes of `OracleUNet.KINDS` set per arm
and compared with the pure-fp32 oracle on
  fwd  : one UNet forward, b = 2 (uncond | text), t = 901: rel L2 of the output
  cfg  : the CFG-combined eps (7.5) of that forward: rel L2
  loop : the final latents of the 10-step DDPM loop of tests/test_gpu_f32.py (tape noise, repellency firing twice)
Writes gpurun_out/precision_ablation.json (copied to profiles/ by hand)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pipeline as opipe            # noqa: E402
from oracle import repellency as orp            # noqa: E402
from oracle import schedulers as osch           # noqa: E402
from oracle.unet import OracleUNet              # noqa: E402
from safe_denoiser_amd.unet import UNet2DConditionModel   # noqa: E402  (synthetic_state_dict only: host code)


def split2(x):
    """two-term bf16 split (hi + lo, 16 mantissa bits): what three bf16 MFMAs (hi*hi + hi*lo + lo*hi) see."""
    hi = x.to(torch.bfloat16).float()
    return hi + (x - hi).to(torch.bfloat16).float()


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm())


def main():
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    steps = int(os.environ.get("ABL_STEPS", "10"))
    sd = UNet2DConditionModel().synthetic_state_dict(1234)
    g = torch.Generator().manual_seed(5)
    E = torch.randn(2, 77, 768, generator=g).to(dev)
    x = torch.randn(1, 4, 64, 64, generator=g).to(dev)
    refs = orp.channel_normalise(torch.randn(64, 4, 64, 64, generator=g)).to(dev)
    tape = torch.randn(40, 1, 4, 64, 64, generator=g).to(dev)
    params = dict(sigma=3.15, scale=0.33, beta_threshold=1e-6, beta_threshold_margin=1e9)

    def run(q_map, act):
        u = OracleUNet(sd, None, act_dtype=act, q_map=q_map, device=dev)
        out = u(torch.cat([x, x]), 901.0, E)
        cfg = out[0:1] + 7.5 * (out[1:2] - out[0:1])
        cur = [0]

        def noise(p, shape):
            z = tape[cur[0]].reshape(shape).clone()
            cur[0] += 1
            return z
        lat, st = opipe.denoise_one(u, osch.DDPM(), E, 0, noise, num_inference_steps=steps,
                                    repel=dict(flavour="threshold", proj_refs=refs, **params))
        return out, cfg, lat, st["renoise_draws"]

    t0 = time.time()
    truth = run(None, None)
    print(f"truth (pure fp32 oracle on {dev}): {time.time() - t0:.1f} s, renoise draws {truth[3]}", flush=True)
    K = OracleUNet.KINDS
    arms = {}
    for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        arms[f"{name}: every class rounded (the engine's 16-bit plan)"] = ({}, dt)
        for k in K:
            arms[f"{name}: all but '{k}' (that class promoted to fp32)"] = ({k: None}, dt)
        for k in K:
            arms[f"{name}: only '{k}' rounded"] = ({kk: (dt if kk == k else None) for kk in K}, None)
        arms[f"{name}: fp32 stream (stream fp32; w/norm/opnd/qkv/ff/inner/text 16-bit)"] = ({"stream": None, "temb": None}, dt)
        arms[f"{name}: MFMA operands only (w, norm, opnd, qkv, ff, text rounded; stream + inner + temb fp32)"] = (
            {"stream": None, "inner": None, "temb": None}, dt)
        arms[f"{name}: MFMA operands only, weights fp32"] = ({"stream": None, "inner": None, "temb": None, "w": None}, dt)
    arms["bf16x2 split operands (hi + lo; 3 MFMAs), fp32 stream"] = (
        {"stream": None, "inner": None, "temb": None, **{k: split2 for k in ("w", "norm", "opnd", "qkv", "ff", "text")}}, None)
    arms["bf16x2 split activations, weights bf16"] = (
        {"stream": None, "inner": None, "temb": None, "w": torch.bfloat16,
         **{k: split2 for k in ("norm", "opnd", "qkv", "ff", "text")}}, None)
    arms["fp16 operands with bf16x2-split weights, fp32 stream"] = (
        {"stream": None, "inner": None, "temb": None, "w": split2,
         **{k: torch.float16 for k in ("norm", "opnd", "qkv", "ff", "text")}}, None)
    res = {"what": __doc__.split("Writes")[0].strip(), "steps": steps, "device": str(dev), "arms": {}}
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    for name, (qm, act) in arms.items():
        t0 = time.time()
        o = run(qm, act)
        res["arms"][name] = {"fwd": rel(o[0], truth[0]), "cfg": rel(o[1], truth[1]), "loop": rel(o[2], truth[2]),
                             "renoise_draws": o[3]}
        print(f"{name:100s} fwd {res['arms'][name]['fwd']:.2e}  cfg {res['arms'][name]['cfg']:.2e}  "
              f"loop {res['arms'][name]['loop']:.2e}  ({time.time() - t0:.1f} s)", flush=True)
        json.dump(res, open(os.path.join(out_dir, "precision_ablation.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
