#!/bin/bash
set -u
export CAMA_VIDEO_SINK=null
O=gpurun_out/r04_demo_loop_cus.txt
: > $O
for e in bgr24 i420; do for n in 0 8 16 32 64; do
  echo "## CAMA_EGRESS=$e CAMA_EGRESS_CUS=$n" >> $O
  CAMA_EGRESS=$e CAMA_EGRESS_CUS=$n timeout 900 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | grep -E "steady state over" >> $O
done; done
cat $O
