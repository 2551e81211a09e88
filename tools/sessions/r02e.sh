#!/bin/bash
# GPU session r02e: the 3:5 raw-frame overlay (k_overlay_raw35): parity tests, bench, kernel trace + SQ PMC.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -x -q -k "raw" > $O/r02e_tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/r02e_tests.log
timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 --cpu-seconds 0 > $O/r02e_raw35.json 2> $O/r02e_raw35.err; echo rc=$?
CAMA_NO_RAW35=1 timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 --cpu-seconds 0 > $O/r02e_rawlds.json 2> $O/r02e_rawlds.err; echo rc=$?
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02e_raw35_trace -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 20 --warmup 3 --cpu-seconds 0 --no-verify > $O/r02e_raw35_trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/r02e_raw35_pmc -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 --cpu-seconds 0 --no-verify > $O/r02e_raw35_pmc.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/r02e_raw35_fetch -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 --cpu-seconds 0 --no-verify > $O/r02e_raw35_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/r02e_raw35_write -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 --cpu-seconds 0 --no-verify > $O/r02e_raw35_write.log 2>&1)
for f in $O/r02e_raw*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step")}, "roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "whole", d["hbm_frac_whole_step"])
PY
done
head -3 $O/r02e_raw35_trace/*/*kernel_stats.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -size +20M -delete
