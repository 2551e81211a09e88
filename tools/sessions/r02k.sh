#!/bin/bash
# GPU session r02k: staged poses + hipGraph replay of the binning chain: parity + benches + host profile.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
B="--cpu-seconds 0"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02k_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02k_tests.log
for v in graph nograph; do
  if [ $v = nograph ]; then export CAMA_NO_GRAPH=1; else unset CAMA_NO_GRAPH; fi
  timeout 300 python bench.py --steps 200 --warmup 10 $B > $O/r02k_head_$v.json 2> $O/r02k_head_$v.err
  timeout 300 python bench.py --height 540 --width 960 --steps 200 --warmup 10 $B > $O/r02k_960_$v.json 2> $O/r02k_960_$v.err
  timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 $B > $O/r02k_raw35_$v.json 2> $O/r02k_raw35_$v.err
  timeout 600 python bench.py --scenes 73 --steps 20 --warmup 2 $B > $O/r02k_scenes73_$v.json 2> $O/r02k_scenes73_$v.err
done
unset CAMA_NO_GRAPH
timeout 300 python tools/host_profile.py --height 540 --width 960 > $O/r02k_host_960.txt 2>&1; head -30 $O/r02k_host_960.txt
for f in $O/r02k_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4), (d.get("hash_check") or {}).get("verified"))
PY
done
