#!/bin/bash
# 24 driver-style headline processes in a row on one box: value, overlay fraction, decided order, launch spread.
set -u
O=gpurun_out/${1:-r04}_process_distribution.txt
: > $O
for i in $(seq 1 ${2:-24}); do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['overlay_mapping']
print(round(d['value']), round(d['sustained']['value']), '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'], m['decided'], [round(x) for x in m['ns_per_mb']], '%.4f..%.4f' % (r['launch_ms_min'], r['launch_ms_max']))" >> $O 2>&1
done
cat $O
