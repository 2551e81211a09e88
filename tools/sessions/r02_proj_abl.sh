#!/bin/bash
# ablation of k_frames_project on the dense 1e6-vertex map: where do its 350 us go?
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
R=$PWD
for v in base ABL_PROJ_NO_CAMS ABL_PROJ_NO_OUT nomask; do
  lib=$R/cama_amd/libcama_hip.so; extra=""
  case $v in ABL*) lib=$R/tools/ab/libcama_$v.so;; nomask) extra="CAMA_NO_CAM_MASK=1";; esac
  (cd /tmp && env CAMA_HIP_LIB=$lib $extra rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_proj_abl_$v -o run -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 --cpu-seconds 0 > $O/r02_proj_abl_$v.json 2>$O/r02_proj_abl_$v.err)
  echo "== $v"; grep -h "k_frames_project\|k_stamps_scatter\|k_overlay\|k_block_cameras" $O/r02_proj_abl_$v/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
