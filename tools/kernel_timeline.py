#!/usr/bin/env python3
"""Timeline of the render chain's kernels from a rocprofv3 --kernel-trace CSV: for the last few overlay launches, when each
kernel of the chain started and ended relative to the overlay before it (do binning and overlay overlap, what does the step
wait for?).

    (cd /tmp && rocprofv3 --output-format csv --kernel-trace -d /tmp/kt -o t -- python bench.py --verts 1000000 --steps 8 ...)
    python tools/kernel_timeline.py /tmp/kt/t_kernel_trace.csv [n_steps]"""
import csv
import sys


def main():
    path = sys.argv[1]
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith("k_") or "fillBuffer" in name:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:28]))
    rows.sort()
    ov = [k for k, r in enumerate(rows) if r[2].startswith("k_overlay")]
    if len(ov) < n_steps + 2:
        print("not enough overlay launches in the trace")
        return
    first = ov[-(n_steps + 1)]
    t0 = rows[first][0]
    print(f"{'kernel':28s} {'start us':>10s} {'end us':>10s} {'dur us':>9s}")
    for s, e, name in rows[first:]:
        print(f"{name:28s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}")
    ovs = [rows[k] for k in ov[-(n_steps + 1):]]
    for a, b in zip(ovs, ovs[1:]):
        print(f"overlay -> next overlay: start to start {(b[0] - a[0]) / 1e3:8.1f} us, gap end -> start {(b[0] - a[1]) / 1e3:7.1f} us")


if __name__ == "__main__":
    main()
