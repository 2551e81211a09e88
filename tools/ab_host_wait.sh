# A/B: pipeline depth x host-side wait for the binning chain (run through gpurun from the repo root; one box = one comparison)
for rep in 1 2; do
for d in 3 2; do
 for hw in -1 1; do
  for cfg in "--height 540 --width 960" "--height 540 --width 960 --raw-frames" "--verts 100000"; do
    export CAMA_PIPELINE_DEPTH=$d CAMA_PIPELINE_HOST_WAIT=$hw
    python bench.py --steps 40 --warmup 10 $cfg --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('rep=$rep depth=$d host_wait=$hw cfg=[$cfg]', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['frac'],3), 'whole', round(d['hbm_frac_whole_step'],3))"
  done
 done
done
done
