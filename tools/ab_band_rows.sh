# band height A/B (profiles/r06_band_rows_ab.txt): the band-height test, then dense / site / headline with the pipeline's own choice (0) and 4 rows forced
python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "band_height or memoised or fullsize_site" 2>&1 | tail -5
for cfg in "--verts 1000000" "--map site --verts 4000000" "--map site --verts 1000000 --sites 3 --scenes 12" ""; do
  for rows in 0 4; do
    echo "== $cfg  forced_rows=$rows"
    CAMA_BENCH_BAND_ROWS=$rows timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras $cfg 2> gpurun_out/bands.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(round(d['value']), 'sustained', round(d['sustained']['value']), 'kernel %.3f whole %.3f sus_whole %.3f' % (r['frac'], d['hbm_frac_whole_step'], d['sustained'].get('hbm_frac_whole_step') or 0), d.get('band_rows',{}).get('last_launch'), d.get('band_rows',{}).get('tall_band_launches'), d['hash_check']['verified'])" || tail -5 gpurun_out/bands.err
  done
done
