#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_jpeg.py -x -q 2>&1 | tail -15
CAMA_JPEG_DIRECT_MAX=0 timeout 900 python -m pytest tests/test_gpu_jpeg.py -x -q -k "restart or fuzz" 2>&1 | tail -3
{ timeout 300 python tools/jpeg_probe.py --batch 240 --restart-rows 1; timeout 300 python tools/jpeg_probe.py --batch 240 --restart-rows 4; timeout 300 python tools/jpeg_probe.py --batch 240; } 2>&1 | grep -v amdgpu.ids | cut -c1-220 > $O/r02_jpeg_dri.txt
cat $O/r02_jpeg_dri.txt
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_jpeg_dri_prof -o run -- python $R/tools/jpeg_probe.py --batch 240 --restart-rows 1 > /dev/null 2>&1)
head -12 $O/r02_jpeg_dri_prof/run_kernel_stats.csv | cut -c1-150
