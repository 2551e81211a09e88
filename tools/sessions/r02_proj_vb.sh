#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
R=$PWD
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -3
CAMA_HIP_LIB=$R/tools/ab/libcama_maskstats.so python bench.py --verts 1000000 --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | grep "mask stats" | head -1
for v in vb8; do
  lib=$R/cama_amd/libcama_hip.so; extra="CAMA_PROJECT_VB=${v#vb}"
  (cd /tmp && env CAMA_HIP_LIB=$lib $extra rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_proj_$v -o run -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 --cpu-seconds 0 > $O/r02_proj_$v.json 2>$O/r02_proj_$v.err)
  echo "== $v $(python -c "import json;d=json.loads([l for l in open('$O/r02_proj_$v.json') if l.startswith('{')][0]);print(round(d['value']), d['ms_per_step'])")"
  grep -h "k_frames_project\|k_stamps_scatter\|k_overlay\|k_block_cameras" $O/r02_proj_$v/run_kernel_stats.csv | python -c "import sys,csv
for r in csv.reader(sys.stdin): print(r[0].split(chr(40))[0][-45:], r[1], round(float(r[3])/1e3,1))"
done
for m in random site; do env python bench.py --verts 1000000 --map $m --steps 30 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('$m', round(d['value']), d['ms_per_step'])"; done
python bench.py --verts 1000000 --steps 30 --warmup 3 --cpu-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('dense', round(d['value']), d['ms_per_step'])"
python bench.py --cpu-seconds 0 2>/dev/null | python -c "import json,sys;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('headline', round(d['value']), d['ms_per_step'], d['hash_check']['verified'])"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_proj_alone -o run -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 --cpu-seconds 0 --no-pipeline > /dev/null 2>&1)
echo "== standalone"; grep -h "k_frames_project\|k_stamps_scatter\|k_overlay\|k_block_cameras" $O/r02_proj_alone/run_kernel_stats.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('  ', r[0].split(chr(40))[0][-45:], r[1], round(float(r[3])/1e3,1))"
