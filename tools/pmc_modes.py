#!/usr/bin/env python3
"""Summarise tools/ubench/run_pmc.sh: per process (one allocation kind, one --pmc set) the overlay's duration and its
counters, for the contiguous (31) and the chunked (5) order.  Each process ran "31:0:0:0,5:0:0:0" with REPS=6: dispatches of
k_overlay 1..9 are order 31 (3 warm-up + 6 timed), 10..18 order 5.

    python tools/pmc_modes.py gpurun_out/pmc_a"""
import collections
import csv
import glob
import os
import sys


def main(root):
    rows = []
    for d in sorted(glob.glob(os.path.join(root, "*_*_*"))):
        if not os.path.isdir(d):
            continue
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            continue
        per = collections.OrderedDict()           # dispatch id -> {counter: value, "us": duration}
        for r in csv.DictReader(open(f[0])):
            if "k_overlay" not in r["Kernel_Name"]:
                continue
            e = per.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp = list(per.values())
        if len(disp) < 18:
            continue
        name = os.path.basename(d)
        proc, alloc = name.split("_")[:2]
        for label, sl in (("31", disp[3:9]), ("5", disp[12:18])):
            keys = [k for k in sl[0] if k != "us"]
            us = sum(x["us"] for x in sl) / len(sl)
            vals = {k: sum(x[k] for x in sl) / len(sl) for k in keys}
            rows.append((proc, alloc, label, us, vals))
    print(f"{'proc':>4} {'alloc':>7} {'order':>5} {'us':>8} {'frac':>6}  counters (mean per launch)")
    for proc, alloc, label, us, vals in rows:
        frac = 2 * 40 * 6 * 900 * 1600 * 3 / (us * 1e-6) / 8e12
        print(f"{proc:>4} {alloc:>7} {label:>5} {us:8.1f} {frac:6.3f}  " + "  ".join(f"{k}={v:.4g}" for k, v in vals.items()))


if __name__ == "__main__":
    main(sys.argv[1])
