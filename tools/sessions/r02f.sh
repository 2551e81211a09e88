#!/bin/bash
# GPU session r02f: project-once binning + LDS-staged raw35: full suite, benches, traces.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02f_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r02f_tests.log
B="--cpu-seconds 0"
timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 $B > $O/r02f_raw35.json 2> $O/r02f_raw35.err; echo rc=$?
timeout 300 python bench.py --steps 100 --warmup 5 $B > $O/r02f_head.json 2> $O/r02f_head.err; echo rc=$?
timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 $B > $O/r02f_n1e5.json 2> $O/r02f_n1e5.err; echo rc=$?
timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02f_dense.json 2> $O/r02f_dense.err; echo rc=$?
timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 $B > $O/r02f_random.json 2> $O/r02f_random.err; echo rc=$?
timeout 300 python bench.py --map site --verts 1000000 --steps 30 --warmup 3 $B > $O/r02f_site.json 2> $O/r02f_site.err; echo rc=$?
timeout 300 python bench.py --height 540 --width 960 --steps 100 --warmup 5 $B > $O/r02f_960.json 2> $O/r02f_960.err; echo rc=$?
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02f_dense_trace -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 $B --no-pipeline --no-verify > $O/r02f_dense_trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/r02f_dense_pmc -- python $R/bench.py --verts 1000000 --steps 4 --warmup 1 $B --no-pipeline --no-verify > $O/r02f_dense_pmc.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02f_raw35_trace -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 20 --warmup 3 $B --no-verify > $O/r02f_raw35_trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/r02f_raw35_pmc -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 $B --no-verify > $O/r02f_raw35_pmc.log 2>&1)
for f in $O/r02f_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4), (d.get("hash_check") or {}).get("verified"))
PY
done
for t in dense raw35; do head -6 $O/r02f_${t}_trace/*/*kernel_stats.csv | cut -c1-150; done
find $O -name "*kernel_trace.csv" -size +20M -delete
