#!/bin/bash
# GPU session r02a: parity of the scalar-camera binning; dense-map A/B (camera matrices via scalar loads vs LDS);
# kernel traces + SQ PMC for the dense binning passes and the raw-frame LDS overlay.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/r02a_tests.log 2>&1
tail -3 $O/r02a_tests.log
for lib in new camlds; do
  if [ $lib = camlds ]; then export CAMA_HIP_LIB=$PWD/tools/ab/libcama_camlds.so; else unset CAMA_HIP_LIB; fi
  timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 --cpu-seconds 0 > $O/r02a_dense_$lib.json 2> $O/r02a_dense_$lib.err
  timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 --cpu-seconds 0 > $O/r02a_n1e5_$lib.json 2> $O/r02a_n1e5_$lib.err
  timeout 300 python bench.py --steps 100 --warmup 5 --cpu-seconds 0 > $O/r02a_head_$lib.json 2> $O/r02a_head_$lib.err
  timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 --cpu-seconds 0 > $O/r02a_random_$lib.json 2> $O/r02a_random_$lib.err
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02a_dense_trace_$lib -- python $PWD/bench.py --verts 1000000 --steps 10 --warmup 2 --cpu-seconds 0 --no-pipeline > $O/r02a_dense_trace_$lib.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/r02a_dense_pmc_$lib -- python $PWD/bench.py --verts 1000000 --steps 4 --warmup 1 --cpu-seconds 0 --no-pipeline > $O/r02a_dense_pmc_$lib.log 2>&1)
done
unset CAMA_HIP_LIB
# raw-frame LDS overlay (reference default 540x960 from raw 1600x900): trace + PMC
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02a_raw_trace -- python $PWD/bench.py --raw-frames --height 540 --width 960 --steps 20 --warmup 3 --cpu-seconds 0 > $O/r02a_raw_trace.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/r02a_raw_pmc -- python $PWD/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 --cpu-seconds 0 > $O/r02a_raw_pmc.log 2>&1)
for f in $O/r02a_*.json; do echo "== $f"; cat $f | python -c "import sys,json; [print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['avg_launch_ms']) for d in map(json.loads, sys.stdin)]"; done
# keep only the small summaries (traces are big)
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*.db" -delete
du -sh $O
