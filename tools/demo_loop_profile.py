"""cProfile of the verbatim main.py loop (tools/demo_loop_probe.py) in steady state."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CAMA_VIDEO_SINK", "null")
import torch
from cama.dataset import ClipManager
from cama.tools import VideoGenerator
from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
root = tempfile.mkdtemp(prefix="cama_demo_")
clip = os.path.join(root, "clip")
N = int(os.environ.get("FRAMES", "240"))
make_clip(clip, n_frames=N + 1, seed=0, n_lines=20, verts_per_line=11, line_len_m=5.0, raster_size=3000,
          image_mode="jpg_photo", image_size=(900, 1600), with_nuscenes=False, extra_labels=False)
cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)


def set_render_ahead(n):
    global cm
    cfg = dict(DEFAULT_CAMA_CONFIGS)
    cfg["render_ahead"] = n
    cm = ClipManager(cfg, clip)


def one_pass():
    vg = VideoGenerator(os.path.join(root, "out.mp4"), (2880, 1080))
    n = 0
    for image_idx, instance_map in cm.yield_frame(dataset="cama"):
        maps_2d_dict = cm.project_all_camera(instance_map)
        image_dict = cm.render_vectors(maps_2d_dict, image_idx)
        image = vg.concate_image(image_dict)
        vg.add_frame(image)
        n += 1
    vg.close()
    return n

for ra in [int(x) for x in os.environ.get("RENDER_AHEAD", "16").split(",")]:
    set_render_ahead(ra)
    one_pass(); one_pass()
    torch.cuda.synchronize()
    for _ in range(3):
        t = time.perf_counter(); n = one_pass(); dt = time.perf_counter() - t
        print(f"render_ahead {ra}: {n} frames in {dt*1e3:.1f} ms = {n/dt:.0f} frames/s")
pr = cProfile.Profile(); pr.enable(); one_pass(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
