#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out
R=$PWD
timeout 600 python -m pytest tests/test_gpu_jpeg.py -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 2>&1 | grep "^photo: 315" | cut -c1-100; done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_jpeg_prof -o run -- python $R/tools/jpeg_probe.py --batch 240 > /dev/null 2>&1)
head -8 $O/r02_jpeg_prof/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
