#!/bin/bash
# clocks / power sampled every ~0.15 s while a bench loop runs for a few seconds:  tools/smi_during.sh <tag> <bench args...>
set -u
tag=$1; shift
O=gpurun_out/smi_$tag.txt
: > $O
python bench.py "$@" --cpu-seconds 0 --sustain-seconds 6 > gpurun_out/smi_${tag}_bench.json 2>/dev/null &
pid=$!
sleep 2
while kill -0 $pid 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|busy" | sed 's/^GPU\[0\][[:space:]]*: //' | tr '\n' ';' >> $O
  echo >> $O
  sleep 0.1
done
wait $pid
python - $O gpurun_out/smi_${tag}_bench.json <<'PY'
import re, sys, json
rows = [l for l in open(sys.argv[1]) if "sclk" in l]
def num(pat, l):
    m = re.search(pat, l); return float(m.group(1)) if m else float("nan")
s = [(num(r"sclk clock level: \d+: \((\d+)Mhz\)", l), num(r"Power \(W\): ([\d.]+)", l), num(r"GPU use \(%\): (\d+)", l)) for l in rows]
busy = [x for x in s if x[2] >= 90]
print(sys.argv[1], len(s), "samples,", len(busy), "with the GPU busy")
if busy:
    sc = sorted(x[0] for x in busy); pw = sorted(x[1] for x in busy)
    print("  sclk MHz min / median / max:", sc[0], sc[len(sc) // 2], sc[-1], "  power W min / median / max:", pw[0], pw[len(pw) // 2], pw[-1])
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("  bench:", round(d["value"]), "frames/s, sustained", round(d["sustained"]["value"]), ", kernel", round(r["frac"], 3), "launch ms min..max", round(r["launch_ms_min"], 4), round(r["launch_ms_max"], 4))
PY
head -3 $O | cut -c1-300
