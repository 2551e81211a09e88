#!/bin/bash
# GPU session r02d: full -m gpu suite (new configs[2]/[4] parity tests), default bench, the N>1 code path on one GPU
# (2 ranks sharing it, gloo) incl. the golden hash check, and the 73-scene sweep on one rank.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02d_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r02d_tests.log
timeout 600 python bench.py > $O/r02d_bench_default.json 2> $O/r02d_bench_default.err; echo "bench rc=$?"
timeout 900 python bench.py --scenes 73 --steps 20 --warmup 2 --cpu-seconds 0 > $O/r02d_bench_scenes73.json 2> $O/r02d_bench_scenes73.err; echo "scenes73 rc=$?"
CAMA_BENCH_SHARE_GPU=1 CAMA_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 > $O/r02d_bench_2ranks.json 2> $O/r02d_bench_2ranks.err; echo "2ranks rc=$?"
tail -3 $O/r02d_bench_2ranks.err
for f in $O/r02d_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step", "scaling", "n_gpus", "hash_check", "per_rank_seconds")})
        print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "whole", d["hbm_frac_whole_step"])
        if "stress" in d: print("stress", {k: d["stress"][k] for k in ("value", "ms_per_step", "hash_check")})
PY
done
