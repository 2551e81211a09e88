#!/bin/bash
# GPU session r02g: ext-launch completion events (A/B), bin stream priority (A/B), pipelined raw35.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
B="--cpu-seconds 0 --no-verify"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_dropin.py -m gpu -x -q > $O/r02g_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02g_tests.log
for rep in 1 2; do
for v in ext noext; do
  if [ $v = noext ]; then export CAMA_NO_EXT_EVENTS=1; else unset CAMA_NO_EXT_EVENTS; fi
  timeout 300 python bench.py --steps 200 --warmup 10 $B > $O/r02g_head_${v}_$rep.json 2> $O/r02g_head_${v}_$rep.err
  timeout 300 python bench.py --height 540 --width 960 --steps 200 --warmup 10 $B > $O/r02g_960_${v}_$rep.json 2>> $O/r02g_head_${v}_$rep.err
done
done
unset CAMA_NO_EXT_EVENTS
timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 $B > $O/r02g_raw35_pipe.json 2> $O/r02g_raw35_pipe.err
timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 $B --no-pipeline > $O/r02g_raw35_nopipe.json 2> $O/r02g_raw35_nopipe.err
for pr in high low; do
  CAMA_BIN_PRIORITY=$pr timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02g_dense_prio_$pr.json 2> $O/r02g_dense_prio_$pr.err
  CAMA_BIN_PRIORITY=$pr timeout 300 python bench.py --steps 100 --warmup 5 $B > $O/r02g_head_prio_$pr.json 2> $O/r02g_head_prio_$pr.err
done
timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02g_dense_prio_none.json 2> $O/r02g_dense_prio_none.err
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02g_head_trace -- python $R/bench.py --steps 30 --warmup 3 $B > $O/r02g_head_trace.log 2>&1)
for f in $O/r02g_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4))
PY
done
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02g_head_trace/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_overlay" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gaps.sort()
print("overlay launches", len(rows), "median gap ns", gaps[len(gaps)//2], "p10", gaps[len(gaps)//10], "p90", gaps[len(gaps)*9//10])
PY
find $O -name "*kernel_trace.csv" -size +20M -delete
