#!/bin/bash
set -u
B=tools/ubench/overlay_modes
O=gpurun_out/modes_${1:-io}.txt
: > $O
run() { echo "## $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; }
S="31:0:0:0:0,31:0:0:0:1,5:0:0:0:0,5:0:0:0:1,0:0:0:0:1,31:0:0:0:0,31:0:0:0:1"
for i in 1 2 3 4 5; do
  for k in malloc contig vmm:512:0; do REPS=12 run $k 40 1 "$S"; done
done
C="31:0:0:1:0,31:0:0:1:1,5:0:0:1:0,5:0:0:1:1"
REPS=12 run malloc 167 4 "$C"
REPS=24 run malloc 40 12 "$C"
grep -c frac $O
