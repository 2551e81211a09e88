#!/bin/bash
# headline with and without pipeline_host_wait, alternating processes
set -u
O=gpurun_out/hostwait_ab.txt
: > $O
for i in 1 2 3 4; do
  for hw in 0 1; do
  CAMA_PIPELINE_HOST_WAIT=$hw timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['overlay_mapping']
print('host_wait $hw', round(d['value']), round(d['sustained']['value']), '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'], m['decided'], '%.4f..%.4f' % (r['launch_ms_min'], r['launch_ms_max']))" >> $O 2>&1
  done
done
cat $O
