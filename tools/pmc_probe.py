"""PMC probe: a known-size streaming copy (calibration) followed by a few whole-scene renders of one bench workload.
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes: TCC slot limit).

    N=1000000 MAP=site F=40 python tools/pmc_probe.py        # env: N, F, H, W, MAP (lanes|random|site), RAW=1, SCENES, SITES

Prints one JSON line (calibration bytes, the workload, cama_bin_stats of one launch) that tools/collect_profiles.py reads."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

E = os.environ.get
a = bench.parse_args([])
a.frames, a.verts, a.height, a.width, a.map = int(E("F", 40)), int(E("N", 10000)), int(E("H", 900)), int(E("W", 1600)), E("MAP", "lanes")
a.raw_frames = E("RAW") == "1"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
a.scenes, a.sites = int(E("SCENES", 1)), int(E("SITES", 0))       # SCENES > 1: every scene rendered once (new buffers per launch)
scenes = [bench.build_scene(a, k, dev) for k in range(a.scenes)]
cm, frames, clip = scenes[0]
from cama_amd import runtime  # noqa: E402
eng = runtime.engine()
rig = cm._rig()
dmap = cm._static("cama").device()
per = max(1, min(a.frames, eng.max_frames_per_call(dmap, rig)))
out = torch.empty(eng.mosaic_shape(rig, per), dtype=torch.uint8, device=dev)
# calibration: elementwise copy of a known byte count (read B, write B)
src = frames[1:1 + min(per, 40)].reshape(-1)
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
for rep in range(5 if a.scenes == 1 else 1):
    for cm_k, _, _ in scenes:
        idx, w2c = cm_k.frame_poses("cama")
        cm_k.render_clip("cama", out=out, poses=(idx[:per], w2c[:per]), frames_per_launch=per)
torch.cuda.synchronize()
stats = None if a.raw_frames else eng.bin_stats()
print(json.dumps({"calib_bytes": src.numel(), "N": dmap.N, "F": per, "W": a.width, "H": a.height, "map": a.map,
                  "raw": a.raw_frames, "scenes": a.scenes, "sites": a.sites, "bin_stats": stats}))
