#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
R=$PWD
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/r02_sq_counters.txt); wc -l $O/r02_sq_counters.txt
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/r02_proj_pmc_$i -- python $R/bench.py --verts 1000000 --steps 3 --warmup 1 --cpu-seconds 0 --no-pipeline > $O/r02_proj_pmc_$i.log 2>&1) || echo "set $i failed"
  python tools/pmc_summary.py $O/r02_proj_pmc_$i k_frames_project k_stamps_scatter 2>/dev/null | cut -c1-120
done
