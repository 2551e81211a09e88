for cfg in "--map random --verts 1000000" "--map site --verts 1000000 --sites 3 --scenes 12"; do
for v in "" NO_RASTER NO_PATCH NO_BOTH ""; do
  if [ -z "$v" ]; then unset CAMA_HIP_LIB; else export CAMA_HIP_LIB=$PWD/tools/ab/libcama_hip_$v.so; fi
  python bench.py $cfg --steps 10 --warmup 3 --cpu-seconds 0 --sustain-seconds 0 --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg'[:24], 'variant ${v:-full}', round(d['value']), '%.4f' % d['ms_per_step'], 'k %.4f ms %.3f' % (r['avg_launch_ms'], r['frac']), (d.get('placement') or {}).get('chosen_ms_mean'))"
done; done
