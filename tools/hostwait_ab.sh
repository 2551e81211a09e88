#!/bin/bash
# pipeline_host_wait off / auto / on, alternating processes; BENCH_ARGS picks the configuration
set -u
O=gpurun_out/hostwait_ab_${1:-headline}.txt
: > $O
for i in 1 2 3; do
  for hw in 0 -1 1; do
  CAMA_PIPELINE_HOST_WAIT=$hw timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; m=d['overlay_mapping']
print('host_wait $hw', round(d['value']), round(d['sustained']['value']), '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'], m['decided'], '%.4f..%.4f' % (r['launch_ms_min'], r['launch_ms_max']), d['hash_check'].get('ok'))" >> $O 2>&1
  done
done
cat $O
