#!/bin/bash
# frames per launch of the 10^6-vertex x 1000-frame stress (BASELINE configs[4] on one GPU), whole step
# and kernel, now that the stamp scratch no longer caps a launch at 167 frames.  gpurun --timeout 2400 -- 'tools/stress_sweep.sh r04'
set -u
tag=${1:-r04}
O=gpurun_out/${tag}_stress_sweep.txt
: > $O
for fpl in 40 80 128 167 250 500 1000; do
  CAMA_FRAMES_PER_LAUNCH=$fpl timeout 900 python bench.py --map random --verts 1000000 --frames 1000 --shard-frames --steps 3 --warmup 1 \
      --cpu-seconds 0 --sustain-seconds 0 > gpurun_out/${tag}_stress_fpl${fpl}.json 2> gpurun_out/${tag}_stress_fpl${fpl}.err
  python - $fpl gpurun_out/${tag}_stress_fpl${fpl}.json >> $O <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(f"fpl {int(sys.argv[1]):5d}  {d['value']:9.0f} frames/s  step {d['ms_per_step']:7.3f} ms  whole-step {d['hbm_frac_whole_step']:.3f}  "
          f"k_overlay {r['frac']:.3f} ({r['launches']} timed launches, {r['avg_launch_ms']:.4f} ms, min {r['launch_ms_min']:.4f} max {r['launch_ms_max']:.4f})  "
          f"scratch {d['scratch_bytes'] / 1e9:.2f} GB  verified {d['hash_check']['verified']}")
except Exception as e:
    print("fpl", sys.argv[1], "FAILED", repr(e))
PY
done
cat $O
