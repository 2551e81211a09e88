#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
B="--cpu-seconds 0 --no-verify --no-pipeline"
for v in base ABL_NO_RASTER ABL_NO_PATCH ABL_NO_STAMPS; do
  if [ $v = base ]; then unset CAMA_HIP_LIB; else export CAMA_HIP_LIB=$R/tools/ab/libcama_$v.so; fi
  timeout 300 python bench.py --verts 1000000 --steps 20 --warmup 3 $B > $O/r02o_dense_$v.json 2> $O/r02o_dense_$v.err
  timeout 300 python bench.py --verts 100000 --steps 30 --warmup 3 $B > $O/r02o_n1e5_$v.json 2>> $O/r02o_dense_$v.err
done
unset CAMA_HIP_LIB
CAMA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $O/r02o_forcedist_nccl.json 2> $O/r02o_forcedist_nccl.err; echo "forcedist rc=$?"; tail -2 $O/r02o_forcedist_nccl.err
for f in $O/r02o_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "overlay ms", d["roofline"]["avg_launch_ms"], d.get("rccl_world"), d.get("collective"), (d.get("hash_check") or {}).get("verified"))
PY
done
