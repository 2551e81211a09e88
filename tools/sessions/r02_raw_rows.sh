#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for r in 2 4 8; do echo "raw rows $r: $(CAMA_BAND_ROWS=$r python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 --cpu-seconds 0 2>&1 | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3))")"; done
for r in 4 8 16; do echo "960x540 rows $r: $(CAMA_BAND_ROWS=$r python bench.py --height 540 --width 960 --steps 200 --warmup 10 --cpu-seconds 0 2>&1 | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3))")"; done
