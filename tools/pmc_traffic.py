#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE; separate passes, TCC has 4 slots) into per-launch HBM
bytes per kernel, with the gfx950 corrections of MI355X_MICROARCH.md section HBM: FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read (x2), WRITE_SIZE is exact for 16-B-per-lane stores; units are KiB.

The record is tied to the code it was collected on: `__meta__` carries the sha256 of the libsdn.so in the tree (what the
profiled process loaded), the git HEAD when known (SDN_GIT_HEAD: the GPU box has no .git) and the batch; bench.py reports
`roofline.traffic` only from a record whose libsdn sha equals the running library's.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [batch]"""
import collections
import csv
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def provenance(batch=None):
    lib = os.path.join(ROOT, "safe_denoiser_amd", "libsdn.so")
    sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
    head = os.environ.get("SDN_GIT_HEAD")
    if not head:
        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
        except Exception:
            head = None
    return {"libsdn_sha256": sha, "git_head": head, "batch": batch,
            "collected_with": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate passes) over bench.py --inference-steps 2"}


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"__meta__": provenance(sys.argv[4] if len(sys.argv) > 4 else None)}
    for k in fetch:
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        out[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * f * 1024.0, "write_bytes_per_launch": w * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
                  "note": "FETCH_SIZE x2 (gfx950 wide-read correction) x1024; WRITE_SIZE x1024"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    rows = {k: v for k, v in out.items() if k != "__meta__"}
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print(f"{k[:70]:70s} x{v['launches']:5d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
