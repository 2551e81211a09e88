#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
B="--cpu-seconds 0 --no-verify"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/r02u_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/r02u_tests.log | cut -c1-200
for v in 256 128 64; do
  if [ $v = 256 ]; then unset CAMA_HIP_LIB; else export CAMA_HIP_LIB=$R/tools/ab/libcama_seg$v.so; fi
  timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02u_dense_$v.json 2>/dev/null
  timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 $B > $O/r02u_random_$v.json 2>/dev/null
  timeout 300 python bench.py --steps 100 --warmup 5 $B > $O/r02u_head_$v.json 2>/dev/null
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02u_trace_$v -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 $B --no-pipeline > /dev/null 2>&1)
  echo "== segs $v"; grep -E "k_stamps" $O/r02u_trace_$v/*/*kernel_stats.csv | cut -d, -f2-4
done
unset CAMA_HIP_LIB
for f in $O/r02u_*.json; do python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1].split("/")[-1], {k: d.get(k) for k in ("value", "ms_per_step")})
PY
done
find $O -name "*kernel_trace.csv" -size +20M -delete
