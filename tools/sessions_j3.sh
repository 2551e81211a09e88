mkdir -p gpurun_out/j3
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/j3/gputest.log 2>&1; echo rc=$? >> gpurun_out/j3/gputest.log
for rep in 1 2; do for q in 4 8 16; do
  echo "== queues=$q" >> gpurun_out/j3/bench_q.txt
  GPU_MAX_HW_QUEUES=$q python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['hbm_frac_whole_step'], d['k_steps_region']['value'], d.get('hw_queues'))" >> gpurun_out/j3/bench_q.txt
done; done
for q in 4 8 16; do
  echo "== queues=$q" >> gpurun_out/j3/demo_q.txt
  GPU_MAX_HW_QUEUES=$q CAMA_VIDEO_SINK=null timeout 300 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | tail -12 >> gpurun_out/j3/demo_q.txt
  echo "== queues=$q nocache" >> gpurun_out/j3/demo_q.txt
  CAMA_FRAME_CACHE_BYTES=0 GPU_MAX_HW_QUEUES=$q CAMA_VIDEO_SINK=null timeout 300 python tools/demo_loop_probe.py --frames 240 --passes 6 2>&1 | tail -12 >> gpurun_out/j3/demo_q.txt
done
python tools/jpeg_probe.py --batch 240 --reps 5 > gpurun_out/j3/jpeg.log 2>&1
tail -3 gpurun_out/j3/gputest.log; cat gpurun_out/j3/bench_q.txt; grep "images/s =" gpurun_out/j3/jpeg.log
