#!/bin/bash
# GPU session r02b: kernel traces + SQ PMC for the dense binning passes (camera matrices via scalar loads vs LDS)
# and the raw-frame LDS overlay.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY"
for lib in new camlds; do
  if [ $lib = camlds ]; then export CAMA_HIP_LIB=$R/tools/ab/libcama_camlds.so; else unset CAMA_HIP_LIB; fi
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02c_dense_trace_$lib -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 --cpu-seconds 0 --no-pipeline > $O/r02c_dense_trace_$lib.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/r02c_dense_pmc_$lib -- python $R/bench.py --verts 1000000 --steps 4 --warmup 1 --cpu-seconds 0 --no-pipeline > $O/r02c_dense_pmc_$lib.log 2>&1)
done
unset CAMA_HIP_LIB
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/r02c_raw_trace -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 20 --warmup 3 --cpu-seconds 0 > $O/r02c_raw_trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/r02c_raw_pmc -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 --cpu-seconds 0 > $O/r02c_raw_pmc.log 2>&1)
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.db" -delete
find $O -path "*r02b*" -name "*stats.csv" | head; du -sh $O
