#!/bin/bash
# what the driver runs at round end: -m gpu suite, smoke(), default bench
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu > $O/r02_verify_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/r02_verify_tests.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $O/r02_verify_bench.json 2> $O/r02_verify_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02_verify_bench.json") if l.startswith("{")][0])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","scaling","dtype")})
print(d["config"]["workload"]); print("roofline", d["roofline"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "hash", d["hash_check"])
PY
