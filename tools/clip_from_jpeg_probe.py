"""Whole-clip render from JPEG files (ClipManager.render_clip: read bytes -> device JPEG decode -> fused raw overlay):
frames/s end to end, against the per-frame main.py loop.  Usage: python tools/clip_from_jpeg_probe.py [--frames 60]"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cama_amd.dataset import ClipManager
from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=60)
a = ap.parse_args()
clip = os.path.join(tempfile.mkdtemp(prefix="cama_clip_"), "clip")
make_clip(clip, n_frames=a.frames + 1, seed=0, n_lines=20, verts_per_line=11, line_len_m=5.0, raster_size=3000,
          image_mode=os.environ.get("CAMA_DEMO_IMAGES", "jpg_photo"), image_size=(900, 1600), with_nuscenes=False, extra_labels=False)
cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
for step in (6, 20, 40):
    for _ in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        idx, out = cm.render_clip("cama", frames_per_launch=step)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print(f"render_clip from {len(idx) * 6} JPEG files, {step} frames per launch: {dt * 1e3:.1f} ms = {len(idx) / dt:.0f} frames/s "
          f"(mosaic {tuple(out.shape)})")
print(cm.frame_source()._jpeg.stats)
