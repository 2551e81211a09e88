# A/B: priority of the pipeline's binning stream on the stress (per-launch spread) and the small-launch workloads, one box
for rep in 1 2; do for pr in 2 1 0; do
  for cfg in "--map random --verts 1000000 --frames 1000 --shard-frames --steps 4 --warmup 2" "--height 540 --width 960 --steps 40 --warmup 10" "--map site --verts 1000000 --scenes 12 --sites 3 --steps 6 --warmup 2"; do
    CAMA_BIN_PRIORITY=$pr python bench.py $cfg --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('prio=$pr cfg=[$cfg]'[:70], round(d['value']), 'kernel', round(r['frac'],3), 'whole', round(d['hbm_frac_whole_step'],3), 'launch min/max', round(r['launch_ms_min'],4), round(r['launch_ms_max'],4), round(r['launch_max_over_min'],3))"
  done
done; done
