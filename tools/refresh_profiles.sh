#!/bin/bash
# Everything under profiles/<tag>_* that comes from bench.py, in one gpurun call (from the repo root, on the GPU box):
#   gpurun --timeout 3000 -- 'tools/refresh_profiles.sh r03'
# then, in the build container:  for n in headline site1e6 random1e6 stress raw35 scenes73 sites3x12; do python tools/collect_profiles.py r06 $n; done
# and copy gpurun_out/<tag>_*_bench.json of the plain lines into profiles/.
set -u
tag=${1:-r06}
ulimit -c 0
tools/profile_workload.sh $tag headline "N=10000" --steps 20 --warmup 5 > /dev/null
tools/profile_workload.sh $tag site1e6 "N=1000000 MAP=site" --map site --verts 1000000 --steps 30 --warmup 5 > /dev/null
tools/profile_workload.sh $tag random1e6 "N=1000000 MAP=random" --map random --verts 1000000 --steps 30 --warmup 5 > /dev/null
tools/profile_workload.sh $tag stress "N=1000000 MAP=random F=1000" --map random --verts 1000000 --frames 1000 --shard-frames --steps 10 --warmup 2 > /dev/null
tools/profile_workload.sh $tag raw35 "N=10000 RAW=1 H=540 W=960" --raw-frames --height 540 --width 960 --steps 100 --warmup 5 > /dev/null
# round 4: rocprof behind every N = 1 BASELINE line (configs[2] = the 73-scene sweep, configs[3] = 3 sites x 12 scenes)
tools/profile_workload.sh $tag scenes73 "N=10000 SCENES=73" --scenes 73 --steps 20 --warmup 2 --cpu-seconds 0 > /dev/null
tools/profile_workload.sh $tag sites3x12 "N=1000000 MAP=site SITES=3 SCENES=12" --map site --verts 1000000 --sites 3 --scenes 12 --steps 10 --warmup 2 --cpu-seconds 0 > /dev/null
# plain bench lines (no trace behind them)
plain() { name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/${tag}_${name}_bench.json 2> gpurun_out/${tag}_${name}_bench.err; tail -c 300 gpurun_out/${tag}_${name}_bench.json; echo; }
plain 960x540 --height 540 --width 960 --steps 60 --warmup 5
plain n1e5 --verts 100000 --steps 30 --warmup 5
plain dense1e6 --verts 1000000 --steps 20 --warmup 3
plain site4e6 --map site --verts 4000000 --steps 20 --warmup 3
plain segments --segments --steps 20 --warmup 5 --cpu-seconds 0
plain wu --segments --wu --steps 20 --warmup 5 --cpu-seconds 0
# N > 1 code path end to end on this one GPU (eight ranks share it, gloo for the one collective): functional, not a measurement
CAMA_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus 8 --steps 2 --warmup 1 > gpurun_out/${tag}_8ranks_on_one_gpu_bench.json 2> gpurun_out/${tag}_8ranks_on_one_gpu_bench.err; tail -c 300 gpurun_out/${tag}_8ranks_on_one_gpu_bench.json; echo
# the reference's real workload: a fresh ClipManager per scene, both passes once (tools/cold_sweep.py)
timeout 600 python tools/cold_sweep.py --scenes 12 --frames 40 --label final --json gpurun_out/${tag}_cold_final.json > gpurun_out/${tag}_cold_final.txt 2>&1
CAMA_FRAME_CACHE_BYTES=0 timeout 600 python tools/cold_sweep.py --scenes 12 --frames 40 --label final_nocache --json gpurun_out/${tag}_cold_final_nocache.json > gpurun_out/${tag}_cold_final_nocache.txt 2>&1
grep -h "^{" gpurun_out/${tag}_cold_final.txt gpurun_out/${tag}_cold_final_nocache.txt | cut -c1-400
# the device JPEG decoder: per-kernel time of a 240-image batch, the decoder-alone rate with the package's eight hardware queues and
# with the HIP runtime's four, and the verbatim main.py loop (bgr24 sink) with and without the decoded-frame cache
bash tools/jpeg_kernel_stats.sh > gpurun_out/${tag}_jpeg_kernel_stats.txt 2>&1
JPEG_SETS=photo bash tools/jpeg_kernel_stats.sh > gpurun_out/${tag}_jpeg_kernel_stats_photo.txt 2>&1
JPEG_SETS=photo bash tools/jpeg_pmc.sh ${tag}_jpmc_photo > gpurun_out/${tag}_jpeg_pmc_photo.txt 2>&1
python tools/jpeg_probe.py --batch 240 --reps 5 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_jpeg_probe.txt
GPU_MAX_HW_QUEUES=4 python tools/jpeg_probe.py --batch 240 --reps 5 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_jpeg_probe_4queues.txt
CAMA_VIDEO_SINK=null timeout 300 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | grep "main.py loop\|steady state over" > gpurun_out/${tag}_demo_loop.txt
CAMA_FRAME_CACHE_BYTES=0 CAMA_VIDEO_SINK=null timeout 300 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | grep "main.py loop\|steady state over" > gpurun_out/${tag}_demo_loop_nocache.txt
grep -h "images/s =\|steady state over" gpurun_out/${tag}_jpeg_probe.txt gpurun_out/${tag}_jpeg_probe_4queues.txt gpurun_out/${tag}_demo_loop.txt gpurun_out/${tag}_demo_loop_nocache.txt
