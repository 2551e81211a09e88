#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
CAMA_FUZZ_ITERS=4000 CAMA_FUZZ_SEED=987654 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
CAMA_FUZZ_ITERS=2000 timeout 3000 python -m pytest tests/test_gpu_jpeg.py -x -q -k fuzz 2>&1 | tail -1
for i in 1 2 3 4 5; do python bench.py --steps 100 --cpu-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['hash_check']['verified'], d['hash_check']['mismatched'])"; done
