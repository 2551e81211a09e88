#!/usr/bin/env python3
"""Launch-to-launch spread of one kernel inside ONE process against its memory-side counters.

    python tools/launch_spread.py <counter_collection.csv> <kernel name prefix>

Input: rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL (or the WRREQ pair): one row per
(dispatch, counter instance) with the dispatch's start / end timestamps.  Under --pmc every dispatch runs alone on the GPU,
so whatever spread is left is the kernel's own -- no neighbour on another stream.  Prints every launch's duration and mean
request latency (LEVEL / REQ, TCC cycles), the fastest and the slowest thirds side by side, and their correlation."""
import collections
import csv
import sys

import numpy as np


def main(path, prefix):
    per = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if not name.startswith(prefix):
            continue
        d = per.setdefault(int(r["Dispatch_Id"]), {"us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = list(per.values())
    if not rows:
        print("no dispatch of", prefix)
        return
    names = [k for k in rows[0] if k != "us"]
    req = next((n for n in names if n.endswith("REQ")), None)
    lvl = next((n for n in names if n.endswith("LEVEL")), None)
    rows = rows[len(rows) // 4:]                               # (drop the warm-up quarter)
    us = np.array([r["us"] for r in rows])
    print(f"{prefix}: {len(us)} launches, duration min {us.min():.1f} / median {np.median(us):.1f} / max {us.max():.1f} us "
          f"(max / min {us.max() / us.min():.3f})")
    if req and lvl:
        lat = np.array([r[lvl] / r[req] for r in rows])
        rq = np.array([r[req] for r in rows])
        order = np.argsort(us)
        k = max(1, len(us) // 3)
        fast, slow = order[:k], order[-k:]
        print(f"  requests per launch {rq.mean():.4g} (spread {rq.std() / rq.mean() * 100:.2f} %)   counters: {req}, {lvl}")
        print(f"  fastest third: {us[fast].mean():7.1f} us, mean latency {lat[fast].mean():7.0f} cycles")
        print(f"  slowest third: {us[slow].mean():7.1f} us, mean latency {lat[slow].mean():7.0f} cycles")
        print(f"  correlation(duration, latency) = {np.corrcoef(us, lat)[0, 1]:.3f};  duration ratio slow / fast "
              f"{us[slow].mean() / us[fast].mean():.3f}, latency ratio {lat[slow].mean() / lat[fast].mean():.3f}")
        print("  launches in issue order (us : latency): " + " ".join(f"{a:.0f}:{b:.0f}" for a, b in zip(us, lat)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
