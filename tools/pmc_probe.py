"""PMC probe: a known-size streaming copy (calibration) followed by a few whole-scene renders.
Run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse, torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
ap = argparse.Namespace(frames=40, verts=int(os.environ.get("N", 10000)), height=int(os.environ.get("H", 900)), width=int(os.environ.get("W", 1600)), map="lanes")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
cm, frames, clip = bench.build_scene(ap, 0, dev)
from cama_amd import runtime
eng = runtime.engine(); rig = cm._rig()
out = torch.empty(eng.mosaic_shape(rig, 40), dtype=torch.uint8, device=dev)
# calibration: elementwise copy of exactly frames[1:] bytes (read B, write B)
a = frames[1:].reshape(-1); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
for _ in range(5): cm.render_clip("cama", out=out)
torch.cuda.synchronize()
print("calib_bytes", a.numel(), "N", cm._static("cama").device().N)
