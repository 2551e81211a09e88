#!/usr/bin/env python3
"""cProfile of the FIRST frame of a new clip in a warm process (what tools/cold_sweep.py reports as `cama first`)."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("CAMA_VIDEO_SINK", "null")
import cold_sweep  # noqa: E402


def main():
    import torch
    from cama.dataset import ClipManager
    from cama.tools import VideoGenerator
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS
    root = tempfile.mkdtemp(prefix="cama_cold_")
    clips = cold_sweep.write_clips(root, 4, 40)
    cm = None
    for k, clip in enumerate(clips):
        if k >= 2:
            pr = cProfile.Profile()
            t0 = time.perf_counter()
            pr.enable()
            cm = None
            pr.disable()
            print(f"clip {k}: dropping the previous ClipManager {1e3 * (time.perf_counter() - t0):.2f} ms")
            pstats.Stats(pr).sort_stats("cumulative").print_stats(20)
        cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
        vg = VideoGenerator(os.path.join(root, "x.mp4"))
        pr = cProfile.Profile() if k != 1 else None          # clip 0: the process's one-off start-up; clips 2, 3: a warm process
        it = cm.yield_frame(dataset="cama")
        t0 = time.perf_counter()
        if pr:
            pr.enable()
        image_idx, instance_map = next(it)
        t1 = time.perf_counter()
        maps_2d_dict = cm.project_all_camera(instance_map)
        t2 = time.perf_counter()
        image_dict = cm.render_vectors(maps_2d_dict, image_idx)
        t3 = time.perf_counter()
        image = vg.concate_image(image_dict)
        t4 = time.perf_counter()
        vg.add_frame(image)
        t5 = time.perf_counter()
        if pr:
            pr.disable()
        print(f"clip {k}: yield {1e3 * (t1 - t0):.2f} project {1e3 * (t2 - t1):.2f} render_vectors {1e3 * (t3 - t2):.2f} "
              f"concate {1e3 * (t4 - t3):.2f} add_frame {1e3 * (t5 - t4):.2f} ms")
        if pr:
            pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
        for image_idx, instance_map in it:
            vg.add_frame(vg.concate_image(cm.render_vectors(cm.project_all_camera(instance_map), image_idx)))
        vg.close()
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
