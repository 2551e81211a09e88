#!/bin/bash
# Does the overlay's speed mode flip in TIME inside one process (same buffers)?  90 s of back-to-back launches, one line per
# ~0.35 s (1000 launches), next to rocm-smi samples of clocks / temperatures / power once a second.
set -u
O=gpurun_out/longrun_${1:-a}
mkdir -p $O
( for i in $(seq 1 100); do date +%s.%N; rocm-smi --showclocks --showtemp --showpower --showmemuse 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Temperature|Power|junction|memory" | tr '\n' ';'; echo; sleep 1; done ) > $O/smi.txt 2>&1 &
SMI=$!
S=""
for i in $(seq 1 260); do S="$S,31:0:0:0"; done
REPS=1000 timeout 200 tools/ubench/overlay_modes ${2:-malloc} 40 1 "${S#,}" | awk '{print systime(), $0}' > $O/runs.txt
kill $SMI 2>/dev/null
awk '{print $1, $NF, $(NF-2)}' $O/runs.txt | awk 'NR%5==0' | head -80
tail -3 $O/smi.txt | cut -c1-600
