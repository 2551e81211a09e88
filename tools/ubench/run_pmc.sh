#!/bin/bash
# gpurun driver: per-channel / per-XCD L2 <-> fabric counters of the overlay in slow (physically contiguous buffers: always)
# and fast / slow (plain hipMalloc: per process) runs.  One --pmc set per pass (TCC slot limit); --kernel-trace only.
set -u
R=$PWD
B=$R/tools/ubench/overlay_modes
O=$R/gpurun_out/pmc_${1:-a}
mkdir -p $O
export TMPDIR=/tmp
SCRIPT="31:0:0:0,5:0:0:0"
i=0
for alloc in contig malloc malloc malloc malloc; do
  i=$((i+1))
  for set in "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL" "TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" "TCC_EA0_WRREQ_STALL TCC_TAG_STALL" "TCC_EA0_RDREQ_GMI_CREDIT_STALL TCC_EA0_WRREQ_GMI_CREDIT_STALL" "TCP_UTCL1_TRANSLATION_MISS GRBM_UTCL2_BUSY" "TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM"; do
    tag=$(echo $set | tr ' ' '+')
    d=$O/${i}_${alloc}_$tag
    (cd /tmp && REPS=6 timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $d -o p -- $B $alloc 40 1 "$SCRIPT" > $d.log 2>&1)
  done
  # the same process shape without counters: which mode is this allocation pattern in right now?
  REPS=12 timeout 120 $B $alloc 40 1 "$SCRIPT" > $O/${i}_${alloc}_plain.log 2>&1
done
find $O -type f ! -name '*counter_collection.csv' ! -name '*kernel_trace.csv' ! -name '*.log' -delete 2>/dev/null
# environment probes
{ echo "## rocm-smi partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | head -30
  echo "## debugfs"; ls /sys/kernel/debug 2>&1 | head; ls /sys/kernel/debug/dri 2>&1 | head
  echo "## kfd topology mem banks"; for f in /sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties; do echo $f; cat $f; done 2>&1 | head -60
  echo "## xnack / hugepage env"; cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/vm_size 2>&1
} > $O/env.txt
du -sh $O; ls $O | head -50
