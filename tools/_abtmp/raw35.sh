run() { echo "== $1"; env $1 timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 200 --warmup 10 --cpu-seconds 0 2> gpurun_out/raw35.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), 'sustained', round(d['sustained']['value']), 'kernel %.3f whole %.3f sus_whole %.3f' % (r['frac'], d['hbm_frac_whole_step'], d['sustained'].get('hbm_frac_whole_step') or 0), 'launch ms %.4f..%.4f ratio %.3f' % (r['launch_ms_min'], r['launch_ms_max'], r['launch_max_over_min']), 'host issue us', round(d['host_issue_us']['mean']))" || tail -5 gpurun_out/raw35.err; }
for rep in 1 2; do
run "A=0"
run "CAMA_PIPELINE_HOST_WAIT=1"
run "CAMA_PIPELINE_DEPTH=3"
run "CAMA_PIPELINE_DEPTH=3 CAMA_PIPELINE_HOST_WAIT=1"
run "CAMA_BIN_PRIORITY=1"
run "CAMA_BIN_PRIORITY=0"
done
echo "== --no-pipeline"; timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 200 --warmup 10 --cpu-seconds 0 --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), 'kernel %.3f whole %.3f' % (r['frac'], d['hbm_frac_whole_step']), 'launch ms %.4f..%.4f' % (r['launch_ms_min'], r['launch_ms_max']))"
