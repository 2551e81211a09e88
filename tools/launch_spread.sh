#!/bin/bash
# tools/launch_spread.sh <tag> <kernel prefix> <bench args...>: the same bench command under two PMC passes (read and write
# request counters of the L2 -> fabric interface), every launch of <kernel prefix> with its duration and latency
set -u
tag=$1; kern=$2; shift 2
args=("$@")
R=$PWD; export TMPDIR=/tmp
run() {
  p=$1; c1=$2; c2=$3
  o=$R/gpurun_out/${tag}_spread_$p
  (cd /tmp && timeout 900 rocprofv3 --output-format csv --kernel-trace --pmc $c1 $c2 -d $o -o s -- python $R/bench.py "${args[@]}" --cpu-seconds 0 --sustain-seconds 0 --no-verify > $o.log 2>&1)
  f=$(find $o -name '*counter_collection.csv' | head -1)
  echo "== $p"; python $R/tools/launch_spread.py "$f" "$kern"
  find $o -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
}
run rd TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL
run wr TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL
