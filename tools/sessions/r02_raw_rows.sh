#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
for v in 10000 100000; do echo "verts $v: $(python bench.py --raw-frames --height 540 --width 960 --verts $v --steps 100 --warmup 5 --cpu-seconds 0 2>&1 | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], round(d['roofline']['frac'],3))")"; done
