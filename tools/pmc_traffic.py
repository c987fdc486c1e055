#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE; separate passes, TCC has 4 slots) into per-launch HBM
bytes per kernel, with the gfx950 corrections of MI355X_MICROARCH.md section HBM: FETCH_SIZE reports exactly 1/2 of
the bytes of a wide coalesced streaming read (x2), WRITE_SIZE is exact for 16-B-per-lane stores; units are KiB.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


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
    out = {}
    for k in fetch:
        f, n = fetch[k]
        w = write.get(k, (0.0, 0))[0]
        out[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * f * 1024.0, "write_bytes_per_launch": w * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
                  "note": "FETCH_SIZE x2 (gfx950 wide-read correction) x1024; WRITE_SIZE x1024"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print(f"{k[:70]:70s} x{v['launches']:5d}  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
