# three builds alternating on one box: tools/ab_lib3.sh a.so b.so c.so -- <bench args>
a=$1; b=$2; c=$3; shift 4
for rep in 1 2; do for lib in $a $b $c; do
  CAMA_HIP_LIB=$PWD/$lib python bench.py "$@" --cpu-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_ms'],4), 'whole', round(d['hbm_frac_whole_step'],3))"
done; done
