#!/bin/bash
# GPU session r02h: per-block camera masks: parity (full suite) + benches + trace.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
B="--cpu-seconds 0"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02h_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02h_tests.log
for v in mask nomask; do
  if [ $v = nomask ]; then export CAMA_NO_CAM_MASK=1; else unset CAMA_NO_CAM_MASK; fi
  timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02h_dense_$v.json 2> $O/r02h_dense_$v.err
  timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 $B > $O/r02h_n1e5_$v.json 2> $O/r02h_n1e5_$v.err
  timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 $B > $O/r02h_random_$v.json 2> $O/r02h_random_$v.err
  timeout 300 python bench.py --map site --verts 1000000 --steps 30 --warmup 3 $B > $O/r02h_site_$v.json 2> $O/r02h_site_$v.err
  timeout 300 python bench.py --map site --verts 4000000 --steps 20 --warmup 3 $B > $O/r02h_site4_$v.json 2> $O/r02h_site4_$v.err
done
unset CAMA_NO_CAM_MASK
timeout 300 python bench.py --steps 100 --warmup 5 $B > $O/r02h_head.json 2> $O/r02h_head.err
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02h_dense_trace -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 $B --no-pipeline --no-verify > $O/r02h_dense_trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02h_random_trace -- python $R/bench.py --map random --verts 1000000 --steps 10 --warmup 2 $B --no-pipeline --no-verify > $O/r02h_random_trace.log 2>&1)
for f in $O/r02h_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4), (d.get("hash_check") or {}).get("verified"))
PY
done
for t in dense random; do grep -E "k_" $O/r02h_${t}_trace/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-110; done
find $O -name "*kernel_trace.csv" -size +20M -delete
