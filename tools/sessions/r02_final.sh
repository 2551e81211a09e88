#!/bin/bash
# GPU session: the measurement set behind DESIGN.md / profiles/ for round 2 (tag r02).
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
ulimit -c 0
R=$PWD
O=$R/gpurun_out
mkdir -p $O
T=r02
timeout 600 python bench.py > $O/${T}_bench_plain.json 2> $O/${T}_bench_plain.err; echo "plain rc=$?"
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $O/${T}_bench_stats -- python $R/bench.py --cpu-seconds 0 > $O/${T}_bench_stats.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/${T}_pmc_fetch -- python $R/tools/pmc_probe.py > $O/${T}_pmc_fetch.log 2>&1)
(cd /tmp && timeout 600 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/${T}_pmc_write -- python $R/tools/pmc_probe.py > $O/${T}_pmc_write.log 2>&1)
B="--cpu-seconds 0"
timeout 300 python bench.py --height 540 --width 960 --steps 200 --warmup 10 $B > $O/${T}_bench_960x540.json 2>/dev/null
timeout 300 python bench.py --raw-frames --height 540 --width 960 --steps 100 --warmup 5 $B > $O/${T}_bench_raw_960x540.json 2>/dev/null
timeout 900 python bench.py --scenes 73 --steps 20 --warmup 2 $B > $O/${T}_bench_scenes73.json 2>/dev/null
timeout 300 python bench.py --verts 100000 --steps 50 --warmup 3 $B > $O/${T}_bench_n1e5.json 2>/dev/null
timeout 300 python bench.py --verts 1000000 --steps 30 --warmup 3 $B > $O/${T}_bench_dense1e6.json 2>/dev/null
timeout 300 python bench.py --map random --verts 1000000 --steps 30 --warmup 3 $B > $O/${T}_bench_random1e6.json 2>/dev/null
timeout 300 python bench.py --map site --verts 1000000 --steps 30 --warmup 3 $B > $O/${T}_bench_site1e6.json 2>/dev/null
timeout 300 python bench.py --map site --verts 4000000 --steps 20 --warmup 3 $B > $O/${T}_bench_site4e6.json 2>/dev/null
timeout 600 python bench.py --map random --verts 1000000 --frames 1000 --steps 10 --warmup 2 $B > $O/${T}_bench_stress_1e6x1000.json 2>/dev/null
CAMA_BENCH_SHARE_GPU=1 CAMA_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 2 > $O/${T}_bench_2ranks_shared_gpu.json 2> $O/${T}_bench_2ranks.err
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/${T}_raw_pmc_fetch -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 $B --no-verify > /dev/null 2>&1)
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/${T}_raw_pmc_write -- python $R/bench.py --raw-frames --height 540 --width 960 --steps 4 --warmup 1 $B --no-verify > /dev/null 2>&1)
timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 > $O/${T}_jpeg_probe.txt 2>&1
timeout 300 python tools/jpeg_probe.py --batch 6 --reps 20 >> $O/${T}_jpeg_probe.txt 2>&1
timeout 300 python tools/jpeg_probe.py --batch 240 --reps 10 --restart-rows 1 2>&1 | grep -v amdgpu.ids > $O/${T}_jpeg_probe_dri.txt
# dense 1e6-vertex map: the binning chain on its own (--no-pipeline: nothing runs beside it), kernel stats + SQ counters
(cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $O/${T}_dense_alone -o run -- python $R/bench.py --verts 1000000 --steps 10 --warmup 2 $B --no-pipeline > /dev/null 2>&1)
cp $O/${T}_dense_alone/run_kernel_stats.csv $O/${T}_dense1e6_standalone_kernel_stats.csv
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --output-format csv --kernel-trace --pmc $PMC -d $O/${T}_dense_pmc_$i -- python $R/bench.py --verts 1000000 --steps 3 --warmup 1 $B --no-pipeline > /dev/null 2>&1)
  python tools/pmc_summary.py $O/${T}_dense_pmc_$i k_frames_project k_stamps_scatter k_block_cameras k_overlay
done > $O/${T}_project_dense1e6_pmc_sq.csv
CAMA_VIDEO_SINK=null timeout 600 python tools/demo_loop_probe.py --frames 240 > $O/${T}_demo_loop.txt 2>&1
timeout 300 python tools/clip_from_jpeg_probe.py > $O/${T}_clip_from_jpeg.txt 2>&1
for f in $O/${T}_bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print({k: d.get(k) for k in ("value", "ms_per_step", "n_gpus")}, "roofline", round(d["roofline"]["frac"],4), d["roofline"]["avg_launch_ms"], "whole", round(d["hbm_frac_whole_step"],4), (d.get("hash_check") or {}).get("verified"), (d.get("cpu_baseline") or {}).get("value"))
        if "stress" in d: print("   stress", d["stress"]["value"], d["stress"]["hash_check"]["verified"])
PY
done
tail -3 $O/${T}_demo_loop.txt; tail -4 $O/${T}_jpeg_probe.txt; tail -3 $O/${T}_clip_from_jpeg.txt
find $O -name "*kernel_trace.csv" -size +20M -delete
