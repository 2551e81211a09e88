#!/bin/bash
# the verbatim main.py loop incl. VideoGenerator, median of 5 steady-state passes, default (bgr24) and
# opt-in I420 egress, next to what the PCIe link allows.
set -u
O=gpurun_out/r04_demo_loop.txt
: > $O
export CAMA_VIDEO_SINK=null
echo "## PCIe" >> $O; python tools/pcie_duplex_probe.py >> $O 2>&1
for e in bgr24 i420; do
  echo "## CAMA_EGRESS=$e" >> $O
  CAMA_EGRESS=$e timeout 900 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | grep -E "loop|steady" >> $O
done
echo "## timeline (bgr24)" >> $O
timeout 900 python tools/loop_timeline.py 2>&1 | tail -40 >> $O
cat $O
