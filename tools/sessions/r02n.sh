#!/bin/bash
# GPU session r02n: read-before-atomic in the stamp rasteriser: parity + dense / headline benches.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
mkdir -p $O
B="--cpu-seconds 0"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -x -q > $O/r02n_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/r02n_tests.log | cut -c1-200
timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/r02n_dense.json 2> $O/r02n_dense.err
timeout 300 python bench.py --steps 200 --warmup 10 $B > $O/r02n_head.json 2> $O/r02n_head.err
timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 $B > $O/r02n_n1e5.json 2> $O/r02n_n1e5.err
timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 $B > $O/r02n_random.json 2> $O/r02n_random.err
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02n_dense_trace -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 $B --no-pipeline --no-verify > $O/r02n_dense_trace.log 2>&1)
for f in $O/r02n_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4), (d.get("hash_check") or {}).get("verified"))
PY
done
grep -E "k_overlay|k_frames|k_stamps" $O/r02n_dense_trace/*/*kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-110
find $O -name "*kernel_trace.csv" -size +20M -delete
