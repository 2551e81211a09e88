#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
CAMA_FUZZ_ITERS=3000 timeout 3000 python -m pytest tests/test_gpu_jpeg.py -x -q -k fuzz 2>&1 | tail -5
