#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_jpeg.py -x -q 2>&1 | tail -2
python tools/demo_loop_profile.py 2>&1 | grep -v "amdgpu.ids\|it/s" | sed -n 1,20p | cut -c1-150
