#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for r in 4 8 16; do echo "rows $r: $(CAMA_BAND_ROWS=$r python bench.py --verts 1000000 --steps 30 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])")"; done
for r in 4 8; do echo "headline rows $r: $(CAMA_BAND_ROWS=$r python bench.py --steps 100 --cpu-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])")"; done
