#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_jpeg.py -x -q 2>&1 | tail -3
{ timeout 300 python tools/jpeg_probe.py --batch 240 --restart-rows 1; } 2>&1 | grep -v amdgpu.ids | cut -c1-220 > $O/r02_jpeg_dri.txt
cat $O/r02_jpeg_dri.txt
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_jpeg_dri_prof -o run -- python $R/tools/jpeg_probe.py --batch 240 --restart-rows 1 > /dev/null 2>&1)
f=$(ls $O/r02_jpeg_dri_prof/*kernel_stats.csv $O/r02_jpeg_dri_prof/*/*kernel_stats.csv 2>/dev/null | head -1); head -14 $f | cut -c1-160
