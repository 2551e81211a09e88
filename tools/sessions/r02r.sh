#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_jpeg.py -m gpu -x -q > $O/r02r_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/r02r_tests.log | cut -c1-300
for v in global lds; do
  if [ $v = lds ]; then export CAMA_HIP_LIB=$R/tools/ab/libcama_jpeg_lds.so; else unset CAMA_HIP_LIB; fi
  for lanes in 4 8; do
    timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 --lanes $lanes > $O/r02r_jpeg_240_${v}_l$lanes.txt 2>&1; echo "== $v lanes=$lanes"; grep "batch 240" $O/r02r_jpeg_240_${v}_l$lanes.txt | cut -c1-110
  done
  timeout 300 python tools/jpeg_probe.py --batch 96 --reps 10 > $O/r02r_jpeg_96_$v.txt 2>&1; grep "batch 96" $O/r02r_jpeg_96_$v.txt | cut -c1-110
  timeout 300 python tools/jpeg_probe.py --batch 6 --reps 20 > $O/r02r_jpeg_6_$v.txt 2>&1; grep "batch 6" $O/r02r_jpeg_6_$v.txt | cut -c1-110
done
unset CAMA_HIP_LIB
CAMA_VIDEO_SINK=null timeout 600 python tools/demo_loop_probe.py --frames 240 > $O/r02r_demo_loop.txt 2>&1; grep "main.py loop" $O/r02r_demo_loop.txt
