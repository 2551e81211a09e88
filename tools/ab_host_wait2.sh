for rep in 1 2; do for hw in -1 1; do for cfg in "--height 540 --width 960" "--height 540 --width 960 --raw-frames" "--verts 100000"; do
CAMA_PIPELINE_HOST_WAIT=$hw python bench.py --steps 60 --warmup 10 $cfg --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('host_wait=$hw cfg=[$cfg]', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'whole', round(d['hbm_frac_whole_step'],3), 'host', round(d['host_issue_us']['mean'],1))"
done; done; done
