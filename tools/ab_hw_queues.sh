# bench workloads x GPU_MAX_HW_QUEUES on one box (alternating): tools/ab_hw_queues.sh "<queue counts>" -- <bench.py args>
qs=$1; shift; [ "$1" = "--" ] && shift
for rep in 1 2; do for q in $qs; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py "$@" --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('queues $q', '$*'[:40], round(d['value']), 'kernel %.3f whole %.3f spread %.3f' % (r['frac'], d['hbm_frac_whole_step'], r.get('launch_max_over_min') or 0))"
done; done
