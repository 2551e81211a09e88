#!/bin/bash
# kernel + memory-copy trace of the verbatim main.py loop (3 passes), to see what the GPU does during a steady pass
set -u
R=$PWD
export TMPDIR=/tmp CAMA_VIDEO_SINK=null
d=$R/gpurun_out/loop_trace_${1:-bgr24}
(cd /tmp && CAMA_EGRESS=${1:-bgr24} timeout 900 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $d -o t -- python $R/tools/demo_loop_probe.py --frames 240 --passes 4 > $d.log 2>&1)
grep -E "steady|loop," $d.log | tail -5
python - $d <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(kt[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][-50:], 0))
for r in csv.DictReader(open(mc[0])) if mc else []:
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Kind", "?")), int(float(r.get("Bytes", r.get("Size", 0)) or 0))))
rows.sort()
t_end = rows[-1][1]
# the last steady pass ~ the last 75 ms
win = [x for x in rows if x[0] >= t_end - 80_000_000]
t0 = win[0][0]
span = (t_end - t0) / 1e6
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for s, e, kind, name, nbytes in win:
    a = agg[(kind, name)]
    a[0] += 1; a[1] += (e - s) / 1e6; a[2] += nbytes
print(f"last {span:.1f} ms of the run:")
for (kind, name), (n, ms, nb) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    extra = f"  {nb / 1e6:9.1f} MB  {nb / ms / 1e6 if ms else 0:6.1f} GB/s" if kind == "C" else ""
    print(f"  {kind} {name:52s} x{n:5d}  {ms:8.2f} ms{extra}")
# union busy time of kernels
iv = sorted((s, e) for s, e, kind, _, _ in win if kind == "K")
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
print(f"  kernels cover {busy / 1e6:.1f} ms of the {span:.1f} ms window")
PY
find $d -type f -name '*.csv' -size +20M -delete 2>/dev/null
