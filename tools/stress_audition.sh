#!/bin/bash
# The 10^6-vertex x 1000-frame stress on one GPU with placed per-launch mosaic buffers against plain allocations.
set -u
O=gpurun_out/${1:-r04}_stress_audition.txt
: > $O
for a in 16 0 16 0; do
  timeout 900 python bench.py --map random --verts 1000000 --frames 1000 --shard-frames --steps 3 --warmup 1 \
      --cpu-seconds 0 --sustain-seconds 0 --audition $a > gpurun_out/stress_a$a.json 2> gpurun_out/stress_a$a.err
  python - $a gpurun_out/stress_a$a.json >> $O <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    r = d["roofline"]; p = d.get("placement") or {}
    print(f"audition {int(sys.argv[1]):3d}  {d['value']:9.0f} frames/s  step {d['ms_per_step']:7.3f} ms  whole-step {d['hbm_frac_whole_step']:.3f}  "
          f"k_overlay {r['frac']:.3f} ({r['launches']} launches, {r['avg_launch_ms']:.4f} ms, min {r['launch_ms_min']:.4f} max {r['launch_ms_max']:.4f})  "
          f"verified {d['hash_check']['verified']}  pool {p.get('first_mosaic_candidates_ms')} kept {p.get('kept_of_pool')}")
except Exception as e:
    print("audition", sys.argv[1], "FAILED", repr(e)); print(open(sys.argv[2].replace('.json', '.err')).read()[-1500:])
PY
done
cat $O
