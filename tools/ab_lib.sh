# A/B of two builds of the library on one box: tools/ab_lib.sh <old.so> <new.so> -- <bench args...>   (alternating, 2 each)
old=$1; new=$2; shift 3
for rep in 1 2; do for lib in $old $new; do
  CAMA_HIP_LIB=$PWD/$lib python bench.py "$@" --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), 'whole', round(d['hbm_frac_whole_step'],3), 'project ms', round(d['roofline_project']['avg_launch_ms'],4), 'stamps', (d.get('projection_stats') or {}).get('stamps'))"
done; done
