#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out
timeout 600 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -2
{ timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10; timeout 300 python tools/jpeg_probe.py --batch 96 --reps 10; timeout 300 python tools/jpeg_probe.py --batch 6 --reps 20; } 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/r02_jpeg_probe.txt
timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 --restart-rows 1 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/r02_jpeg_probe_dri.txt
for l in 16 20 24 32; do echo "noise lanes $l: $(timeout 300 python tools/jpeg_probe.py --batch 240 --lanes $l --min-group 4 2>&1 | grep '^noise: 1298' | cut -c1-75)"; done
CAMA_VIDEO_SINK=null timeout 600 python tools/demo_loop_probe.py --frames 120 2>&1 | tail -3 | tee $O/r02_demo_loop.txt
timeout 300 python tools/clip_from_jpeg_probe.py 2>&1 | tail -4 | tee $O/r02_clip_from_jpeg.txt
