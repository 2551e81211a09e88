#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_dropin.py -m gpu -x -q > $O/r02q_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02q_tests.log | cut -c1-300
timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 > $O/r02q_jpeg_240.txt 2>&1; tail -6 $O/r02q_jpeg_240.txt
timeout 300 python tools/jpeg_probe.py --batch 96 --reps 10 > $O/r02q_jpeg_96.txt 2>&1; tail -6 $O/r02q_jpeg_96.txt
CAMA_VIDEO_SINK=null timeout 600 python tools/demo_loop_probe.py --frames 240 > $O/r02q_demo_loop.txt 2>&1; grep "main.py loop" $O/r02q_demo_loop.txt
timeout 300 python tools/clip_from_jpeg_probe.py > $O/r02q_clip_from_jpeg.txt 2>&1; tail -4 $O/r02q_clip_from_jpeg.txt
