#!/bin/bash
# HIP API calls per step of the 960x540 pipelined bench: this tree against the round-3 tree (_ab/old)
set -u
export TMPDIR=/tmp
R=$PWD
for which in new old; do
  d=$R; [ $which = old ] && d=$R/_ab/old
  o=$R/gpurun_out/hipcalls_$which
  (cd /tmp && rocprofv3 --output-format csv --hip-runtime-trace --stats -d $o -o t -- python $d/bench.py --height 540 --width 960 --steps 200 --warmup 5 --cpu-seconds 0 --sustain-seconds 0 --no-verify ${EXTRA:-} > $o.log 2>&1)
  echo "== $which"; python - $o <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*hip_api_stats.csv", recursive=True) or glob.glob(sys.argv[1] + "/**/*api_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:14]:
    print(f"  {r['Name']:40s} calls {int(r['Calls']):7d}  per step {int(r['Calls']) / 205:6.2f}  avg {float(r['AverageNs']) / 1e3:7.2f} us")
PY
done
