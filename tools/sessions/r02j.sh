#!/bin/bash
# GPU session r02j: host-side profile of a pipelined step; JPEG decoder baseline + kernel trace.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout 300 python tools/host_profile.py --height 540 --width 960 > $O/r02j_host_960.txt 2>&1; head -45 $O/r02j_host_960.txt
timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 > $O/r02j_jpeg_240.txt 2>&1; tail -4 $O/r02j_jpeg_240.txt
timeout 300 python tools/jpeg_probe.py --batch 96 --reps 10 > $O/r02j_jpeg_96.txt 2>&1; tail -4 $O/r02j_jpeg_96.txt
timeout 300 python tools/jpeg_probe.py --batch 240 --reps 5 --restart-rows 1 > $O/r02j_jpeg_240_dri.txt 2>&1; tail -4 $O/r02j_jpeg_240_dri.txt
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02j_jpeg_trace -- python $R/tools/jpeg_probe.py --batch 240 --reps 5 > $O/r02j_jpeg_trace.log 2>&1)
grep -E "k_jpeg" $O/r02j_jpeg_trace/*/*kernel_stats.csv | cut -d, -f1-8 | sed 's/(anonymous namespace):://g' | cut -c1-150
find $O -name "*kernel_trace.csv" -size +20M -delete
