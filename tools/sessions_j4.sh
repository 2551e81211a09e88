mkdir -p gpurun_out/j4
for q in 8 16; do for g in 1 2 3 0; do for d in 3 4; do
  echo "== queues=$q pump_groups=$g ahead=$d nocache" >> gpurun_out/j4/demo.txt
  CAMA_DECODE_AHEAD=$d CAMA_PUMP_GROUPS=$g CAMA_FRAME_CACHE_BYTES=0 GPU_MAX_HW_QUEUES=$q CAMA_VIDEO_SINK=null timeout 300 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | grep "steady state over" >> gpurun_out/j4/demo.txt
done; done; done
timeout 600 python tools/cold_sweep.py --scenes 12 --frames 40 --label q8 --json gpurun_out/j4/cold_q8.json > gpurun_out/j4/cold_q8.txt 2>&1
GPU_MAX_HW_QUEUES=4 timeout 600 python tools/cold_sweep.py --scenes 12 --frames 40 --label q4 --json gpurun_out/j4/cold_q4.json > gpurun_out/j4/cold_q4.txt 2>&1
cat gpurun_out/j4/demo.txt | paste - -  | cut -c1-200
tail -5 gpurun_out/j4/cold_q8.txt | cut -c1-400; tail -5 gpurun_out/j4/cold_q4.txt | cut -c1-400
