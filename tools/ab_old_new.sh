#!/bin/bash
# A/B of the round-3 tree (git worktree under _ab/old) against the working tree, alternating, on one box.
set -u
O=$PWD/gpurun_out/ab_old_new.txt
: > $O
for rep in 1 2 3; do
  for which in old new; do
    d=$PWD; [ $which = old ] && d=$PWD/_ab/old
    for cfg in "--height 540 --width 960 --steps 60 --warmup 5" "--steps 20 --warmup 5" "--scenes 24 --steps 10 --warmup 2"; do
      (cd $d && timeout 600 python bench.py $cfg --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$which', '$cfg'.split()[0:4], round(d['value']), round(d.get('sustained',{}).get('value',0)), round(d['roofline']['frac'],3), d.get('overlay_mapping',{}).get('decided'))") >> $O 2>&1
    done
  done
done
cat $O
