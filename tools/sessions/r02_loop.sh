#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q 2>&1 | tail -1
CAMA_VIDEO_SINK=null timeout 900 python tools/demo_loop_probe.py --frames 240 2>&1 | grep "main.py loop"
timeout 600 python examples/demo_synthetic.py --frames 24 2>&1 | grep "main.py loop"
