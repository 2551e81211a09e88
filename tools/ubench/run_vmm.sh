#!/bin/bash
set -u
B=tools/ubench/overlay_modes
O=gpurun_out/modes_${1:-vmm}.txt
mkdir -p gpurun_out
: > $O
run() { echo "## $*" >> $O; timeout 300 $B "$@" >> $O 2>&1; }
S="31:0:0:0,5:0:0:0,0:0:0:0,31:0:0:0"
for i in 1 2 3; do
  for k in vmm:2:0 vmm:2:1 vmm:2:2 vmm:32:0 vmm:32:1 vmm:256:0 vmm:256:1 vmm:2048:0 malloc contig; do REPS=12 run $k 40 1 "$S"; done
done
# per-XCD / per-channel read latency: json output keeps the dimensions
export TMPDIR=/tmp
R=$PWD
for alloc in contig malloc malloc vmm:2:1; do
  d=$R/gpurun_out/pmcj_$(echo $alloc | tr ':' '_')_$RANDOM
  (cd /tmp && REPS=4 timeout 300 rocprofv3 --output-format json --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL -d $d -o p -- $R/$B $alloc 40 1 "31:0:0:0" > $d.log 2>&1)
done
ls -la gpurun_out/pmcj_* | head -20
grep -c frac $O
