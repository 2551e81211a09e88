#!/bin/bash
# A/B: JPEG subsequence size x workgroup size (occupancy of the entropy stages)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
for lib in tools/ab/libcama_jpeg_s6_w128.so tools/ab/libcama_jpeg_s6_w256.so tools/ab/libcama_jpeg_s5_w128.so; do
  echo "== $lib"
  CAMA_HIP_LIB=$PWD/$lib timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q 2>&1 | tail -1
  CAMA_HIP_LIB=$PWD/$lib timeout 300 python tools/jpeg_probe.py --batch 240 2>&1 | grep -v amdgpu.ids | cut -c1-200
done > $O/r02_jpeg_ab.txt 2>&1
cat $O/r02_jpeg_ab.txt
