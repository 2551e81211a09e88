#!/bin/bash
# Fast / slow processes: bench.py under rocprofv3 --kernel-trace, several processes in a row; per process the overlay's mean
# duration, the chain kernels' durations, when the chain starts relative to its overlay, and the queues the kernels ran on.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/mode_probe.txt
for P in 1 2 3 4 5 6 7 8; do
    (cd /tmp && rm -rf /tmp/mp && timeout 300 rocprofv3 --output-format csv --kernel-trace -d /tmp/mp -o m -- python $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --sustain-seconds 0 --no-verify > /tmp/mp.json 2>/tmp/mp.err)
    python - "$P" >> gpurun_out/mode_probe.txt <<'PY'
import csv, sys, glob, json, collections
tr = glob.glob("/tmp/mp/**/m_kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list); q = collections.defaultdict(set)
ov = []
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:24]
    if not n.startswith("k_"): continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n].append((e - s) / 1e3); q[n].add(r.get("Queue_Id"))
    if n.startswith("k_overlay"): ov.append((s, e))
    if n.startswith("k_frames_project") and ov: dur["project_start_after_overlay_start"].append((s - ov[-1][0]) / 1e3)
try: b = json.loads(open("/tmp/mp.json").read()); v = (round(b["value"]), round(b["roofline"]["frac"], 3))
except Exception as ex: v = repr(ex)
print("process", sys.argv[1], v, {k: round(sum(x[-15:]) / len(x[-15:]), 1) for k, x in dur.items()}, {k: sorted(x) for k, x in q.items() if k.startswith(("k_overlay", "k_frames"))})
PY
done
cat gpurun_out/mode_probe.txt
