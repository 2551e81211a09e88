#!/usr/bin/env python3
"""main.py's scene loop (main.py:21-70) driven by a config.yaml, on this build's drop-in classes.

    python examples/run_config.py -c examples/config.yaml --synthetic [--frames 8] [--sink null]

  --synthetic   write a synthetic clip (JPEG camera frames, CAMA + nuScenes labels, poses, calibration) for every scene
                name in the config that has no converted clip yet -- neither box has nuScenes data
  --sink        where the raw video stream goes when there is no ffmpeg: "null", or a directory that receives
                <scene>_<dataset>.bgr24 files (CAMA_VIDEO_SINK does the same for an unchanged main.py)

The loop body is the reference's, verbatim; `Reprojector` (cama.reproject) is the same path as one object and is used for
the second dataset pass to show both spellings.
"""
import argparse
import os
import sys
import time
import zipfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def extract_dir_from_zip(zip_filepath, dir_in_zip, dest_dir):
    with zipfile.ZipFile(zip_filepath, "r") as z:
        for member in z.namelist():
            if member.startswith(dir_in_zip):
                z.extract(member, dest_dir)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Read a configuration file.")
    ap.add_argument("-c", "--config", type=str, default=os.path.join(os.path.dirname(__file__), "config.yaml"))
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--sink", default=None)
    ap.add_argument("--root", default=None, help="override converted_dataroot / output_video_dir (tests)")
    args = ap.parse_args(argv)
    from cama.dataset import ClipManager
    from cama.reproject import Reprojector, load_configs
    from cama.tools import VideoGenerator
    from dataset.nuscenes2clip import nuScenes2Clip
    configs = load_configs(args.config)                          # yaml.safe_load + the contract's keys
    if args.root:
        configs["converted_dataroot"] = os.path.join(args.root, "clips")
        configs["output_video_dir"] = os.path.join(args.root, "videos")
    output_dir = configs["converted_dataroot"]
    os.makedirs(output_dir, exist_ok=True)
    if args.synthetic:
        from cama_amd.synth import make_clip
        for k, scene_name in enumerate(configs["scene_names"]):
            clip = os.path.join(output_dir, scene_name)
            if not os.path.exists(os.path.join(clip, "attribute.json")):
                make_clip(clip, n_frames=args.frames + 1, seed=k, n_lines=20, verts_per_line=11, line_len_m=5.0,
                          raster_size=3000, image_mode="jpg_photo", image_size=(900, 1600))
    s2c = nuScenes2Clip(configs)
    done = []
    for scene_name in configs["scene_names"]:
        s2c.convert(scene_name)                                  # (this build: checks that the converted clip exists)
        zip_file = configs["cama_label_file"]
        if os.path.exists(zip_file):
            extract_dir_from_zip(zip_file, f"{scene_name}/", output_dir)
        output_video_dir = configs["output_video_dir"]
        clip_path = os.path.join(output_dir, scene_name)
        os.makedirs(output_video_dir, exist_ok=True)

        def video(name):
            path = os.path.join(output_video_dir, name)
            if args.sink is None:
                return VideoGenerator(path)
            sink = open(os.devnull if args.sink == "null" else os.path.join(args.sink, name + ".bgr24"), "wb")
            return VideoGenerator(path, sink=sink)

        t0 = time.perf_counter()
        cm = ClipManager(configs["cama_configs"], clip_path)
        n = 0
        print("Generating reprojection video with CAMA labels...")
        vg = video(f"{scene_name}_cama.mp4")
        for image_idx, instance_map in cm.yield_frame(dataset="cama"):
            maps_2d_dict = cm.project_all_camera(instance_map)
            image_dict = cm.render_vectors(maps_2d_dict, image_idx)
            image = vg.concate_image(image_dict)
            vg.add_frame(image)
            n += 1
        vg.close()
        print("Generating reprojection video with nuScenes labels...")
        rp = Reprojector(configs, clip_path)                     # the same path as one object
        vg = video(f"{scene_name}_nuScenes.mp4")
        for image_idx, mosaic in rp.frames("nuscenes"):
            vg.add_frame(mosaic)
            n += 1
        vg.close()
        dt = time.perf_counter() - t0
        done.append((scene_name, n, dt))
        print(f"{scene_name}: {n} frames in {dt:.2f} s")
    return done


if __name__ == "__main__":
    main()
