import os, sys, time, tempfile, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from cama.dataset import ClipManager
from cama.tools import VideoGenerator
from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
root = tempfile.mkdtemp(prefix="cama_demo_")
clip = os.path.join(root, "clip")
make_clip(clip, n_frames=61, seed=0, n_lines=20, verts_per_line=11, line_len_m=5.0, raster_size=3000,
          image_mode=os.environ.get("CAMA_DEMO_IMAGES", "jpg_photo"), image_size=(900, 1600), with_nuscenes=False, extra_labels=False)
cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
vg = object.__new__(VideoGenerator)
def loop():
    n = 0
    for image_idx, instance_map in cm.yield_frame(dataset="cama"):
        maps_2d_dict = cm.project_all_camera(instance_map)
        image_dict = cm.render_vectors(maps_2d_dict, image_idx)
        image = vg.concate_image(image_dict)
        memoryview(np.ascontiguousarray(image, dtype=np.uint8)).cast("B")      # what VideoGenerator.add_frame pipes out
        n += 1
    return n
loop()
torch.cuda.synchronize(); t = time.perf_counter(); n = loop(); torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"steady loop: {n} frames in {dt:.3f} s = {n/dt:.1f} fps")
pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(28)
