#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for lib in cama_amd/libcama_hip.so tools/ab/libcama_ov128.so tools/ab/libcama_ov128u5.so; do
  echo "$lib headline: $(CAMA_HIP_LIB=$PWD/$lib python bench.py --steps 100 --cpu-seconds 0 2>&1 | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'], (d.get('hash_check') or {}).get('verified'))")"
  echo "$lib dense: $(CAMA_HIP_LIB=$PWD/$lib python bench.py --verts 1000000 --steps 30 --warmup 3 --cpu-seconds 0 2>&1 | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'], d['roofline']['avg_launch_ms'])")"
done
