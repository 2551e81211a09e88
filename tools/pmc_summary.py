#!/usr/bin/env python3
"""Summarise a rocprofv3 `--pmc` counter_collection.csv per kernel: mean of every counter over the kernel's
dispatches and its ratio to SQ_WAVE_CYCLES.

    python tools/pmc_summary.py gpurun_out/<run dir> [kernel-substring ...] > profiles/<name>.csv
"""
import collections
import csv
import glob
import os
import sys


def main():
    run = sys.argv[1]
    pats = sys.argv[2:] or ["k_"]
    path = max(glob.glob(os.path.join(run, "*", "*counter_collection.csv")) + glob.glob(os.path.join(run, "*counter_collection.csv")),
               key=os.path.getmtime)                   # the newest run (gpurun merges sessions into one directory)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if any(p in k for p in pats):
            name = k.replace("(anonymous namespace)::", "").replace("void ", "")
            per[name.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatches", "counter", "mean_per_dispatch", "ratio_to_SQ_WAVE_CYCLES"])
    for k, c in per.items():
        wc = sum(c.get("SQ_WAVE_CYCLES", [0])) / max(1, len(c.get("SQ_WAVE_CYCLES", [0])))
        for n, v in sorted(c.items()):
            m = sum(v) / len(v)
            w.writerow([k, len(v), n, f"{m:.0f}", f"{m / wc:.4f}" if wc else ""])


if __name__ == "__main__":
    main()
