#!/bin/bash
# GPU session r02i: overlay occupancy vs overlap with the binning stream (LDS pad A/B); egress / render-ahead tests.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
B="--cpu-seconds 0 --no-verify"
for pad in 0 6400 14000; do
  export CAMA_OVERLAY_LDS_PAD=$pad
  timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02i_dense_pad$pad.json 2> $O/r02i_dense_pad$pad.err
  timeout 300 python bench.py --steps 100 --warmup 5 $B > $O/r02i_head_pad$pad.json 2> $O/r02i_head_pad$pad.err
  timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 $B > $O/r02i_n1e5_pad$pad.json 2> $O/r02i_n1e5_pad$pad.err
done
unset CAMA_OVERLAY_LDS_PAD
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q > $O/r02i_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r02i_tests.log
CAMA_VIDEO_SINK=null timeout 600 python tools/demo_loop_probe.py > $O/r02i_demo_loop.txt 2>&1; tail -12 $O/r02i_demo_loop.txt
for f in $O/r02i_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4))
PY
done
