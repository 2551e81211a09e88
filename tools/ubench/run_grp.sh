#!/bin/bash
set -u
B=tools/ubench/overlay_modes
O=gpurun_out/modes_${1:-grp}.txt
: > $O
run() { echo "## $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; }
S="31:0:0:0:0:0,5:0:0:0:0:0,5:0:0:0:0:2,5:0:0:0:0:1,3:0:0:0:0:2,7:0:0:0:0:2,9:0:0:0:0:2,0:0:0:0:0:2,7:0:0:0:0:1,9:0:0:0:0:1,31:0:0:0:0:0"
for i in 1 2 3 4; do
  for k in malloc contig; do REPS=12 run $k 40 1 "$S"; done
done
C="31:0:0:1:0:0,5:0:0:1:0:0,5:0:0:1:0:2,7:0:0:1:0:2,5:0:0:1:0:1"
REPS=12 run malloc 167 4 "$C"
REPS=24 run malloc 40 12 "$C"
grep -c frac $O
