python -m pytest tests -m gpu -x -q -k "site or guard or stress or planned or pipeline or thread" 2>&1 | tail -3
for i in 1 2; do python bench.py --map site --verts 1000000 --sites 3 --scenes 12 --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('sites3x12', round(d['value']), '%.3f' % d['ms_per_step'], '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'], '%.4f' % r['avg_launch_ms'], d['overlay_mapping']['decided'], d['hash_check']['verified'])"; done
python bench.py --map random --verts 1000000 --frames 1000 --shard-frames --steps 10 --warmup 2 --cpu-seconds 0 --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('stress', round(d['value']), '%.3f' % d['ms_per_step'], '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'], d['hash_check']['verified'])"
python bench.py --map site --verts 4000000 --steps 10 --warmup 2 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('site4e6', round(d['value']), '%.3f' % d['ms_per_step'], '%.3f' % r['frac'], '%.3f' % d['hbm_frac_whole_step'])"
