#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from rocprofv3 --pmc passes over tools/unet_forward.py (north star: "MFMA utilisation vs
gfx950 peak").  Inputs: the counter_collection.csv of
   pass 1:  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
   pass 2:  SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
(dispatch durations under the profiler come from the Start / End timestamps in pass 1's rows).
Derivations (MI355X_MICROARCH.md, "Per-instruction cycle constants" / DVFS give-back; checked in round 2 against the MFMA count of
a known tile): SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs in shader cycles; SQ_BUSY_CYCLES is summed over the 32
SQ instances (8 XCDs x 4 shader engines), so elapsed shader cycles = SQ_BUSY_CYCLES / 32;
   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x elapsed cycles);  clock = GRBM_GUI_ACTIVE / 8 / duration (8 XCDs);
SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES (both quad-cycles) = share of wave time stalled on LDS issue.
The record carries the libsdn.so sha256 like the traffic record; bench.py reports roofline.mfma_busy only from a matching one.
usage: pmc_mfma.py <pass1 counter csv> <pass2 counter csv> <out.json> [top=8]"""
import collections
import csv
import json
import sys

from pmc_traffic import provenance


def counters(path):
    """per kernel: counter sums, dispatch count, and the summed dispatch durations (every row of counter_collection.csv carries
    its dispatch's Start / End timestamps: one dispatch appears once per counter)."""
    tot = collections.defaultdict(lambda: collections.Counter())
    cnt, dur = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r.get("Dispatch_Id"))
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
            dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return tot, cnt, dur


def main():
    c1, n1, dur = counters(sys.argv[1])
    nd = n1
    c2, n2, _ = counters(sys.argv[2])
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    out = {"__meta__": dict(provenance("3 x 64 (UNet only, tools/unet_forward.py)"),
                            collected_with="rocprofv3 --pmc, two passes over tools/unet_forward.py (UNet forwards only); durations from pass 1's kernel trace")}
    rows = []
    for k in c1:
        a = c1[k]
        if a.get("SQ_BUSY_CYCLES", 0) <= 0 or k not in dur:
            continue
        elapsed = a["SQ_BUSY_CYCLES"] / 32.0
        ns = dur[k]
        b = c2.get(k, {})
        rec = {"launches": n1[k], "avg_launch_us": ns / max(nd[k], 1) / 1e3, "time_share_ms": ns / 1e6,
               "mfma_busy": a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * elapsed),
               "clock_ghz_sq": elapsed / ns if ns else None,
               "clock_ghz_grbm": (a["GRBM_GUI_ACTIVE"] / 8.0 / ns) if a.get("GRBM_GUI_ACTIVE") and ns else None,
               "lds_issue_stall_share_of_wave_cycles": (b["SQ_WAIT_INST_LDS"] / b["SQ_WAVE_CYCLES"]) if b.get("SQ_WAVE_CYCLES") else None,
               "issue_stall_share_of_wave_cycles": (b["SQ_WAIT_INST_ANY"] / b["SQ_WAVE_CYCLES"]) if b.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in b else None,
               "valu_insts_per_launch": (b["SQ_INSTS_VALU"] / max(n2[k], 1)) if "SQ_INSTS_VALU" in b else None}
        rows.append((ns, k, rec))
    rows.sort(reverse=True)
    for ns, k, rec in rows[:top]:
        out[k] = rec
        print(f"{k[:64]:64s} x{rec['launches']:4d} {rec['avg_launch_us']:9.1f} us  MFMA busy {100 * rec['mfma_busy']:5.1f} %  "
              f"clock {rec['clock_ghz_sq'] or 0:.2f} GHz  LDS-issue stall {100 * (rec['lds_issue_stall_share_of_wave_cycles'] or 0):4.1f} %")
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
