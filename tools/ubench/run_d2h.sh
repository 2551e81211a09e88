#!/bin/bash
set -u
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/d2h
mkdir -p $O
AMD_LOG_LEVEL=4 python $GRAFT_REPO_ROOT/tools/ubench/d2h_torch.py hip 2> $O/log_hip.txt | grep GB/s
grep -i "copy\|sdma\|blit" $O/log_hip.txt | grep -v "hipMemcpyAsync (" | sort | uniq -c | sort -rn | head -12 | cut -c1-260
echo "--- preload system runtime"
for m in hip torch; do
LD_PRELOAD=/opt/rocm/lib/libamdhip64.so:/opt/rocm/lib/libhsa-runtime64.so rocprofv3 --output-format csv --kernel-trace --memory-copy-trace --stats -d $O/pre_$m -o t -- python $GRAFT_REPO_ROOT/tools/ubench/d2h_torch.py $m 2>&1 | grep "GB/s\|rror" | head -3
echo "  [$m preload] blit kernels: $(grep -c copyBuffer $O/pre_$m/t_kernel_trace.csv 2>/dev/null)  $(cut -d, -f1-3 $O/pre_$m/t_memory_copy_stats.csv 2>/dev/null | tail -2 | tr '\n' ' ')"
done
env | grep -i "^HSA\|^HIP\|^GPU_\|^ROC" | head
