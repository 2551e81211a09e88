#!/bin/bash
# Kernel timeline of the 10^6-vertex x 1000-frame stress (one GPU): what the GPU does between two overlay launches.
set -u
R=$PWD
export TMPDIR=/tmp
d=$R/gpurun_out/${1:-stress}_trace
(cd /tmp && timeout 900 rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d $d -o t -- python ${BENCH_PY:-$R/bench.py} ${BENCH_ARGS:---map random --verts 1000000 \
   --frames 1000 --shard-frames --steps 3 --warmup 1} --cpu-seconds 0 --sustain-seconds 0 > $d.log 2>&1)
tail -c 300 $d.log | head -c 200; echo
python - $d ${SHOW:-17} <<'PY'
import csv, glob, re, sys
d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)
rows = []
for r in csv.DictReader(open(kt[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), (re.search(r"(k_\w+)", r["Kernel_Name"]) or re.search(r"(\w+)\W*$", r["Kernel_Name"].split("<")[0])).group(1)[-40:], r.get("Queue_Id", "?")))
for r in csv.DictReader(open(mc[0])) if mc else []:
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")[-20:], "-"))
rows.sort()
ov = [i for i, r in enumerate(rows) if "k_overlay" in r[2]]
# the last 16 overlay launches = the last two timed steps
ov = ov[-int(sys.argv[2]):]
t0 = rows[ov[0]][0]
print("timeline (us from the first shown overlay launch), queue, kernel, duration:")
for i in range(ov[0], ov[-1] + 1):
    s, e, n, q = rows[i]
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f}  q{q:>3}  {n:40s} {(e - s) / 1e3:9.1f}")
gaps = [(rows[b][0] - rows[a][1]) / 1e3 for a, b in zip(ov, ov[1:])]
print("gaps between consecutive overlay launches (us):", [round(g, 1) for g in gaps])
PY
