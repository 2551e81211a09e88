#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
CAMA_FUZZ_ITERS=6000 timeout 3000 python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5
