# A/B of the pipeline's scratch-slot count on the host-/chain-paced workloads (run through gpurun from the repo root)
for d in 3 2; do
  for cfg in "--height 540 --width 960" "--verts 100000" "" "--height 540 --width 960 --raw-frames" "--map site --verts 1000000 --scenes 4 --sites 1"; do
    export CAMA_PIPELINE_DEPTH=$d
    python bench.py --steps 40 --warmup 10 $cfg --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('depth=$d cfg=[$cfg]', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['frac'],3), 'whole', round(d['hbm_frac_whole_step'],3), 'scratch MB', d['scratch_bytes']>>20)"
  done
done
