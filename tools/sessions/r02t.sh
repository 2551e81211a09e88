#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
timeout 600 python bench.py --cpu-seconds 0 > $O/r02t_head.json 2> $O/r02t_head.err; echo rc=$?
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02t_bench_stats -- python $R/bench.py --cpu-seconds 0 > $O/r02t_bench_stats.log 2>&1)
timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 --cpu-seconds 0 > $O/r02t_raw.json 2>/dev/null
for f in $O/r02t_head.json $O/r02t_bench_stats.log $O/r02t_raw.json; do python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(sys.argv[1].split("/")[-1], {k: d.get(k) for k in ("value", "ms_per_step")}, "live kernel ms", d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], "frac", round(d["roofline"]["frac"],4), "whole", round(d["hbm_frac_whole_step"],4))
PY
done
grep -E "k_overlay" $O/r02t_bench_stats/*/*kernel_stats.csv | cut -d, -f2-5 | cut -c1-120
