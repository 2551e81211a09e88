#!/bin/bash
# gpurun driver for tools/ubench/overlay_modes (round 4): see profiles/r04_overlay_modes.txt for what came out.
set -u
B=tools/ubench/overlay_modes
O=gpurun_out/modes_${1:-a}.txt
mkdir -p gpurun_out
: > $O
run() { echo "## $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; }
WARM="31:0:0:0,5:0:0:0,0:0:0:0,31:0:0:0,31:1:0:0,31:7:0:0,31:53:0:0,31:211:0:0,31:845:0:0,31:3375:0:0,31:0:32:0,31:0:128:0,31:0:512:0,31:0:0:0"
for i in 1 2 3 4; do run malloc 40 1 "$WARM"; done
for i in 1 2 3; do run contig 40 1 "$WARM"; done
COLD="31:0:0:1,5:0:0:1,0:0:0:1,31:0:32:1,31:0:64:1,31:0:128:1,31:0:256:1,31:0:512:1,31:0:1024:1,31:0:2048:1,5:0:128:1,5:0:512:1,31:0:0:1,31:211:0:1,31:211:256:1"
for i in 1 2; do run malloc 40 12 "$COLD"; done
run arena 40 12 "$COLD"
run contig 40 12 "$COLD"
BIG="31:0:0:1,5:0:0:1,31:0:128:1,31:0:512:1,31:0:2048:1,31:0:0:1"
REPS=12 run malloc 167 4 "$BIG"
REPS=12 run malloc 80 8 "$BIG"
(cd /tmp && rocprofv3 -L > $OLDPWD/gpurun_out/counters_list.txt 2>&1 || rocprofv3 --list-avail > $OLDPWD/gpurun_out/counters_list.txt 2>&1)
grep -c . gpurun_out/counters_list.txt
tail -5 $O
