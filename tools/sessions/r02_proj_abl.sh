#!/bin/bash
# ablation of k_frames_project on the dense 1e6-vertex map, standalone (--no-pipeline: nothing runs beside it)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
O=$PWD/gpurun_out
R=$PWD
for v in base ABL_PROJ_NO_CAMS ABL_PROJ_NO_OUT; do
  lib=$R/cama_amd/libcama_hip.so
  case $v in ABL*) lib=$R/tools/ab/libcama_$v.so;; esac
  (cd /tmp && env CAMA_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_proj_abl_$v -o run -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 --cpu-seconds 0 --no-pipeline > $O/r02_proj_abl_$v.json 2>$O/r02_proj_abl_$v.err)
  echo "== $v"; grep -h "k_frames_project\|k_stamps_scatter\|k_overlay\|k_block_cameras" $O/r02_proj_abl_$v/run_kernel_stats.csv | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('  ', r[0].split(chr(40))[0][-45:], r[1], round(float(r[3])/1e3,1))"
done 2>&1 | tee $O/r02_project_ablation.txt
