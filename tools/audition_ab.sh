#!/bin/bash
# Headline processes alternating plain allocations (--audition 0) and auditioned ones, one box: value, overlay fraction,
# whole-step fraction, decided order, launch spread, and what the audition saw.
set -u
O=gpurun_out/${1:-r04}_audition_ab.txt
: > $O
for i in $(seq 1 ${2:-8}); do
  for a in 0 16; do
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --audition $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['overlay_mapping']; p=d.get('placement')
print('audition $a', round(d['value']), round(d['sustained']['value']), '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'], m['decided'], '%.4f..%.4f' % (r['launch_ms_min'], r['launch_ms_max']), (p['first_mosaic_candidates_ms'], p['first_frames_candidates_ms']) if p else '')" >> $O 2>&1
  done
done
cat $O
