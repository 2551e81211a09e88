for cfg in "--segments" "--segments --wu" "--raw-frames --height 540 --width 960 --unfused-resample" "--no-pipeline" "--map random --verts 200000 --frames 48 --shard-frames --height 180 --width 320" "--scenes 3 --no-scene-batch" "--map site --verts 300000 --sites 2 --scenes 4 --frames 8" "--audition 0" "--height 450 --width 800 --raw-frames"; do
  python bench.py --steps 6 --warmup 2 $cfg --cpu-seconds 0 --sustain-seconds 0.2 > /tmp/v.json 2> /tmp/v.err; rc=$?
  python - "$cfg" $rc <<'PY'
import json,sys
cfg,rc=sys.argv[1],sys.argv[2]
try:
    d=json.loads(open('/tmp/v.json').read().strip().splitlines()[-1])
    print("rc",rc,"cfg=[%s]"%cfg, round(d["value"]), "whole", round(d["hbm_frac_whole_step"],3), "hash", d["hash_check"]["verified"] if d.get("hash_check") else None, "placement", d["placement"]["source"][:14], "unplaced" in d["placement"], "nomemo" , "without_memo" in d)
except Exception as e:
    print("rc",rc,"cfg=[%s]"%cfg,"NO LINE",repr(e)); print(open('/tmp/v.err').read()[-800:])
PY
done
