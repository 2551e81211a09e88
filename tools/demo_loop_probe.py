#!/usr/bin/env python3
"""The reference's main.py loop (main.py:56-61), verbatim, on a synthetic clip with 1600x900 JPEG frames, VideoGenerator
included: frames/s end to end (files -> device JPEG decode -> fused raw overlay -> pinned download of the bgr24 mosaics [default:
the reference's stream] or device I420 + download [CAMA_EGRESS=i420] -> sink).

    CAMA_VIDEO_SINK=null python tools/demo_loop_probe.py [--frames 240] [--content photo|noise] [--passes 6]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--content", choices=["noise", "photo"], default="photo")
    ap.add_argument("--passes", type=int, default=6, help="passes over the clip: the first is one-off setup, the rest steady state")
    ap.add_argument("--distinct", action="store_true",
                    help="write every JPEG of the clip separately (Pillow: ~0.1 s per file, minutes for 240 frames) instead of "
                         "hard-linking 8 distinct frames per camera over the clip's timestamps (same decode work per file)")
    ap.add_argument("--sweep", default="", help="NAME=v1,v2,...: repeat the steady-state measurement on the same clip with the "
                                                "environment variable NAME set to each value (a fresh ClipManager each)")
    args = ap.parse_args()
    import torch
    from cama.dataset import ClipManager
    from cama.tools import VideoGenerator
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
    root = tempfile.mkdtemp(prefix="cama_demo_")
    clip = os.path.join(root, "clip")
    t = time.perf_counter()
    if args.distinct or args.content == "noise":
        make_clip(clip, n_frames=args.frames + 1, seed=0, n_lines=20, verts_per_line=11, line_len_m=5.0, raster_size=3000,
                  image_mode="jpg" if args.content == "noise" else "jpg_photo", image_size=(900, 1600), with_nuscenes=False,
                  extra_labels=False)
    else:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from loop_timeline import fast_jpeg_clip
        fast_jpeg_clip(clip, args.frames + 1)
    print(f"clip with {6 * (args.frames + 1)} JPEGs written in {time.perf_counter() - t:.1f} s")
    settings = [None]
    if args.sweep:
        if ";" in args.sweep or args.sweep.count("=") > 1:          # "A=1;B=2,A=3;B=4": several variables per setting
            settings = [[kv.split("=") for kv in one.split(";")] for one in args.sweep.split(",")]
        else:
            name, vals = args.sweep.split("=")
            settings = [[(name, v)] for v in vals.split(",")]
    for setting in settings:
        if setting is not None:
            for k, v in setting:
                os.environ[k] = v
            print("## " + " ".join(f"{k}={v}" for k, v in setting))
        run_passes(args, root, clip, ClipManager, VideoGenerator, DEFAULT_CAMA_CONFIGS, torch)


def run_passes(args, root, clip, ClipManager, VideoGenerator, DEFAULT_CAMA_CONFIGS, torch):
    # reference default output size (540, 960); PROBE_RENDER_AHEAD: frames per render batch (configs["render_ahead"], 16)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, render_ahead=int(os.environ.get("PROBE_RENDER_AHEAD", "16"))), clip)
    rates = []
    for label in ["first pass (one-off setup)"] + ["steady state"] * (args.passes - 1):
        vg = VideoGenerator(os.path.join(root, "out.mp4"), (2880, 1080))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        # ---- main.py:57-61 ----
        for image_idx, instance_map in cm.yield_frame(dataset="cama"):
            maps_2d_dict = cm.project_all_camera(instance_map)
            image_dict = cm.render_vectors(maps_2d_dict, image_idx)
            image = vg.concate_image(image_dict)
            vg.add_frame(image)
            n += 1
        # ------------------------
        vg.close()
        dt = time.perf_counter() - t0
        print(f"main.py loop, {label}: {n} frames in {dt:.3f} s = {n / dt:.1f} frames/s (stream: {vg.pix_fmt}, "
              f"mosaic {tuple(image.shape)}, {type(image).__name__})")
        if label == "steady state":
            rates.append(n / dt)
    if rates:
        rates.sort()
        print(f"steady state over {len(rates)} passes: median {rates[len(rates) // 2]:.0f} frames/s (min {rates[0]:.0f}, max {rates[-1]:.0f}); "
              f"egress format {os.environ.get('CAMA_EGRESS', 'bgr24 (default)')}")


if __name__ == "__main__":
    main()
