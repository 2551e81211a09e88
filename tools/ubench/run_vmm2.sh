#!/bin/bash
set -u
B=tools/ubench/overlay_modes
O=gpurun_out/modes_${1:-vmm2}.txt
: > $O
run() { echo "## $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; }
S="31:0:0:0,5:0:0:0,31:0:0:0"
for i in 1 2 3 4; do
  for k in vmm:64:0 vmm:128:0 vmm:256:0 vmm:512:0 vmm:1024:0 vmm:256:0:256 vmm:256:1:256 vmm:1024:0:1024 vmm:2048:0:2048 vmm:128:1 vmm:512:1 malloc; do REPS=12 run $k 40 1 "$S"; done
done
grep -c frac $O
