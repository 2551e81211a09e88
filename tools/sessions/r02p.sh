#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
B="--cpu-seconds 0"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02p_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02p_tests.log | cut -c1-300
timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02p_dense.json 2> $O/r02p_dense.err
timeout 300 python bench.py --verts 1000000 --steps 20 --warmup 3 $B --no-pipeline --no-verify > $O/r02p_dense_nopipe.json 2>> $O/r02p_dense.err
timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 $B > $O/r02p_n1e5.json 2>> $O/r02p_dense.err
timeout 300 python bench.py --steps 200 --warmup 10 $B > $O/r02p_head.json 2>> $O/r02p_dense.err
timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 $B > $O/r02p_random.json 2>> $O/r02p_dense.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 CAMA_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 > $O/r02p_forcedist_nccl.json 2> $O/r02p_forcedist_nccl.err; echo "forcedist rc=$?"; tail -2 $O/r02p_forcedist_nccl.err
for f in $O/r02p_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "overlay ms", d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4), d.get("rccl_world"), d.get("collective"), (d.get("hash_check") or {}).get("verified"))
PY
done
