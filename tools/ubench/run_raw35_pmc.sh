#!/bin/bash
# Issue-cycle / latency account of the 3:5 raw overlay: SQ counters of the classic kernel and of the
# wave-specialised one, one --pmc set per pass.
set -u
R=$PWD
B=$R/tools/ubench/raw35_modes
O=$R/gpurun_out/raw35_pmc
mkdir -p $O
export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_SALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_CYCLES" \
           "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL" "TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL"; do
  tag=$(echo $set | tr ' ' '+')
  (cd /tmp && REPS=6 timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $O/$tag -o p -- $B 40 "0:-1:1,2:-1:3" > $O/$tag.log 2>&1)
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*.log' -delete 2>/dev/null
ls $O | head -30
