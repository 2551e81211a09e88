#!/usr/bin/env python3
"""Per-XCD / per-L2-channel read latency of the overlay from a `rocprofv3 --output-format json --pmc TCC_EA0_RDREQ
TCC_EA0_RDREQ_LEVEL` run (tools/ubench/run_vmm.sh): latency = RDREQ_LEVEL / RDREQ in TCC cycles, per XCD (its 16 channels
summed) and the spread over the 16 channels inside each XCD.

    python tools/pmc_xcd_latency.py gpurun_out/pmcj_*/p_results.json"""
import json
import sys

import numpy as np


def main(paths):
    for p in paths:
        r = json.load(open(p))["rocprofiler-sdk-tool"][0]
        names = {c["id"]["handle"]: c["name"] for c in r["counters"]}
        ksym = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in r["kernel_symbols"]}
        rows = []
        for rec in r["callback_records"]["counter_collection"]:
            di = rec["dispatch_data"]["dispatch_info"]
            if "k_overlay" not in ksym.get(di["kernel_id"], ""):
                continue
            us = (rec["dispatch_data"]["end_timestamp"] - rec["dispatch_data"]["start_timestamp"]) / 1e3
            vals = {}
            for x in rec["records"]:
                vals.setdefault(names[x["counter_id"]["handle"]], []).append(x["value"])
            if len(vals.get("TCC_EA0_RDREQ", [])) != 128:
                continue
            req = np.asarray(vals["TCC_EA0_RDREQ"]).reshape(8, 16)          # [xcc][channel] (instance list order)
            lvl = np.asarray(vals["TCC_EA0_RDREQ_LEVEL"]).reshape(8, 16)
            rows.append((us, req, lvl))
        if not rows:
            print(p, "no overlay dispatches")
            continue
        rows = rows[-4:]                                                     # the timed launches
        us = np.mean([x[0] for x in rows])
        req = np.mean([x[1] for x in rows], axis=0)
        lvl = np.mean([x[2] for x in rows], axis=0)
        lat_x = lvl.sum(1) / req.sum(1)
        lat_c = lvl / req
        print(f"{p}\n  overlay {us:.1f} us   frac {2 * 40 * 6 * 900 * 1600 * 3 / (us * 1e-6) / 8e12:.3f}   mean latency {lvl.sum() / req.sum():.0f} cycles")
        print("  per XCD  requests(k): " + " ".join(f"{v / 1e3:7.1f}" for v in req.sum(1)))
        print("  per XCD  latency:     " + " ".join(f"{v:7.0f}" for v in lat_x))
        print("  per XCD  channel latency min..max: " + " ".join(f"{lat_c[x].min():.0f}..{lat_c[x].max():.0f}" for x in range(8)))
        print("  per channel (all XCDs summed) latency: " + " ".join(f"{v:.0f}" for v in lvl.sum(0) / req.sum(0)))


if __name__ == "__main__":
    main(sys.argv[1:])
