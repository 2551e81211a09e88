for p in default high low; do
  for cfg in "--height 540 --width 960" "--verts 100000" ""; do
    if [ $p = default ]; then unset CAMA_BIN_PRIORITY; else export CAMA_BIN_PRIORITY=$p; fi
    python bench.py --steps 40 --warmup 10 $cfg --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('prio=$p cfg=[$cfg]', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['frac'],3), 'whole', round(d['hbm_frac_whole_step'],3))"
  done
done
