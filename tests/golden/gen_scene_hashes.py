#!/usr/bin/env python3
"""Golden per-scene / per-frame mosaic hashes of bench.py's fixed workloads, rendered by the ORACLE on the CPU.

    python tests/golden/gen_scene_hashes.py sweep   [--scenes 73] [--jobs 8]     # BASELINE configs[1]/[2]
    python tests/golden/gen_scene_hashes.py stress                                # BASELINE configs[4], sampled frames
    python tests/golden/gen_scene_hashes.py small                                 # reduced-size twins for -m gpu tests
    python tests/golden/gen_scene_hashes.py site-full                             # BASELINE configs[3] at full size (12 scenes)
    python tests/golden/gen_scene_hashes.py custom --bench-args "--map random --verts 1000000"   # any other bench.py line

Writes / updates tests/golden/scene_hashes.json: {workload key: {unit id: [lo hex, hi hex]}} where a unit is a scene id
(whole-scene workloads: hash over the scene's [F, 2H, 3W, 3] mosaics) or a frame position (frame-sharded stress: hash of
that frame's mosaic).  Everything here is numpy + oracle/ (the checker); nothing from cama_amd's device path runs, so the
numbers are independent of the product: bench.py --gpus N and tests/test_gpu_configs.py compare the GPU's hashes with them.
The clips, maps and camera-frame bytes come from the same seeded generators bench.py uses (bench.write_scene_clip,
bench.replace_map, synth.frame_pattern_np)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
OUT = os.path.join(HERE, "scene_hashes.json")


_RAW_MAPS = {}                                          # id(cams) -> per-camera (mapx, mapy): one scene's cameras at a time


def scene_setup(bargs, seed):
    """(xyz, colour, cams, [w2c per rendered frame]) of scene `seed`, all through the oracle."""
    import bench
    from oracle import cama_oracle as O
    from cama_amd.synth import CAMERA_NAMES, DEFAULT_CAMA_CONFIGS
    tmp = tempfile.mkdtemp(prefix=f"cama_gold_s{seed}_")
    clip = os.path.join(tmp, "clip")
    bench.write_scene_clip(bargs, seed, clip)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(bargs.height, bargs.width)) for n in CAMERA_NAMES]
    if bargs.map == "lanes":
        labels = json.load(open(os.path.join(clip, "maps", "map_labels.json")))
        bev = np.load(os.path.join(clip, "maps", "vision_road_mlp_ft.npy"))
        static = O.static_map_cama(bev, labels)
    else:
        class Holder:                                   # bench.replace_map fills instance_maps["cama"]
            instance_maps = {}
        bench.replace_map(Holder, bargs, seed)
        static = Holder.instance_maps["cama"]
    xyz, col, counts, _ = O.flatten_instances(static)
    if getattr(bargs, "segments", False) or getattr(bargs, "wu", False):
        # the segment extensions: every point but the first of its instance is joined to its predecessor
        link = np.concatenate([np.arange(n) > 0 for n in counts]) if len(counts) else np.zeros(0, bool)
        col = (col, link)
    stamps, poses = O.pose_track(clip, att, dict(DEFAULT_CAMA_CONFIGS), "cama")
    secs = O.sensor_seconds(att, "camera_front")
    w2c = [O.frame_world2chassis(stamps, poses, secs[i]) for i in range(1, len(secs))]
    assert len(w2c) == bargs.frames
    return xyz, col, cams, w2c


def render_frame(bargs, seed, pos, xyz, col, cams, w2c):
    """Oracle mosaic [2H, 3W, 3] of rendered position `pos` (image index pos + 1) of scene `seed`."""
    from oracle import cama_oracle as O
    from cama_amd.synth import frame_pattern_np
    H, W = bargs.height, bargs.width
    if getattr(bargs, "raw_frames", False):
        # bench.py --raw-frames: 1600x900 sensor frames, undistorted + resized per camera (cama/reproject.py:232-240) before drawing
        raw = frame_pattern_np(seed, (6, 900, 1600, 3), first=(pos + 1) * 6 * 900 * 1600 * 3)
        maps = _RAW_MAPS.setdefault(id(cams), [O.undistort_map(c["K_origin"], c["d_origin"], c["K"], W, H) for c in cams])
        src = np.stack([O.remap_bilinear(raw[c], maps[c][0], maps[c][1]) for c in range(6)])
    else:
        src = frame_pattern_np(seed, (6, H, W, 3), first=(pos + 1) * 6 * H * W * 3)
    flat = O.frame_project_flat(xyz, w2c[pos], cams, W, H)
    if isinstance(col, tuple):                          # (colour, link): bench.py --segments [--wu]
        fn = O.frame_render_flat_wu if getattr(bargs, "wu", False) else O.frame_render_flat_segments
        return fn(src, flat["vu"], flat["vis"], col[0], col[1])
    return O.frame_render_flat(src, flat["vu"], flat["vis"], col)


def scene_hash(job):
    bargs, seed = job
    from cama_amd.shard import overlay_hash_np
    t0 = time.time()
    xyz, col, cams, w2c = scene_setup(bargs, seed)
    H, W = bargs.height, bargs.width
    mos = np.empty((bargs.frames, 2 * H, 3 * W, 3), np.uint8)
    for pos in range(bargs.frames):
        mos[pos] = render_frame(bargs, seed, pos, xyz, col, cams, w2c)
    lo, hi = overlay_hash_np(mos)
    print(f"scene {seed}: {len(xyz)} verts, {time.time() - t0:.1f} s", flush=True)
    return seed, "%016x" % lo, "%016x" % hi


def frame_hashes(bargs, seed, positions):
    from cama_amd.shard import overlay_hash_np
    xyz, col, cams, w2c = scene_setup(bargs, seed)
    out = {}
    for pos in positions:
        t0 = time.time()
        lo, hi = overlay_hash_np(render_frame(bargs, seed, pos, xyz, col, cams, w2c))
        out[str(pos)] = ["%016x" % lo, "%016x" % hi]
        print(f"frame {pos}: {time.time() - t0:.1f} s", flush=True)
    return out


def update(key, units):
    rec = json.load(open(OUT)) if os.path.exists(OUT) else {}
    rec.setdefault(key, {}).update(units)
    rec[key] = {k: rec[key][k] for k in sorted(rec[key], key=int)}
    json.dump(rec, open(OUT, "w"), indent=0, sort_keys=True)
    print(f"{OUT}: {key}: {len(rec[key])} units")


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["sweep", "stress", "small", "site", "site-full", "custom"])
    ap.add_argument("--bench-args", default="", help='custom: the bench.py arguments of the workload, e.g. "--map random --verts 1000000"')
    ap.add_argument("--scenes", type=int, default=bench.SWEEP_SCENES)
    ap.add_argument("--jobs", type=int, default=min(8, os.cpu_count() or 1))
    a = ap.parse_args()
    if a.what == "sweep":
        bargs = bench.parse_args([])
        with mp.get_context("spawn").Pool(a.jobs) as pool:
            res = pool.map(scene_hash, [(bargs, s) for s in range(a.scenes)], chunksize=1)
        update(bench.workload_key(bargs.frames, bargs.verts, bargs.width, bargs.height, "lanes"),
               {str(s): [lo, hi] for s, lo, hi in res})
    elif a.what == "site":
        site_twin(a.jobs)
    elif a.what == "custom":
        # any other whole-scene workload of bench.py (its non-BASELINE lines: big maps, other sizes): every scene's hash
        bargs = bench.parse_args(a.bench_args.split())
        n = max(1, bargs.scenes)
        with mp.get_context("spawn").Pool(min(a.jobs, n)) as pool:
            res = pool.map(scene_hash, [(bargs, s) for s in range(n)], chunksize=1)
        update(bench.args_key(bargs), {str(s): [lo, hi] for s, lo, hi in res})
    elif a.what == "site-full":
        # BASELINE configs[3] as bench.py runs it on one GPU: 12 scenes over three 10^6-vertex site maps, 40 frames at 1600x900
        bargs = bench.parse_args(["--map", "site", "--verts", "1000000", "--sites", "3", "--scenes", "12"])
        with mp.get_context("spawn").Pool(a.jobs) as pool:
            res = pool.map(scene_hash, [(bargs, s) for s in range(12)], chunksize=1)
        update(bench.args_key(bargs), {str(s): [lo, hi] for s, lo, hi in res})
    elif a.what == "stress":
        bargs = bench.parse_args([])
        bargs.map, bargs.verts, bargs.frames = "random", bench.STRESS["verts"], bench.STRESS["frames"]
        units = frame_hashes(bargs, 0, bench.stress_sample_frames(bargs.frames))
        update(bench.workload_key(bargs.frames, bargs.verts, bargs.width, bargs.height, "random", unit="frame"), units)
    else:
        # reduced-size twins used by tests/test_gpu_configs.py: 24 small scenes; a 1e6-vertex random map at 320x180
        bargs = bench.parse_args(["--frames", "6", "--verts", "3000", "--height", "180", "--width", "320"])
        with mp.get_context("spawn").Pool(a.jobs) as pool:
            res = pool.map(scene_hash, [(bargs, s) for s in range(24)], chunksize=1)
        update(bench.workload_key(6, 3000, 320, 180, "lanes"), {str(s): [lo, hi] for s, lo, hi in res})
        bargs = bench.parse_args(["--frames", "125", "--verts", "1000000", "--height", "180", "--width", "320", "--map", "random"])
        units = frame_hashes(bargs, 0, [0, 1, 31, 62, 93, 124])
        update(bench.workload_key(125, 1000000, 320, 180, "random", unit="frame"), units)
        site_twin(a.jobs)


def site_twin(jobs):
    """configs[3] twin for tests/test_gpu_configs.py: 2 sites of 60 000 vertices, 8 scenes (scene k on site k % 2, its own
    drive), 6 frames at 320x180."""
    import bench
    bargs = bench.parse_args(["--frames", "6", "--verts", "60000", "--height", "180", "--width", "320", "--map", "site",
                              "--sites", "2", "--scenes", "8"])
    with mp.get_context("spawn").Pool(jobs) as pool:
        res = pool.map(scene_hash, [(bargs, s) for s in range(8)], chunksize=1)
    update(bench.args_key(bargs), {str(s): [lo, hi] for s, lo, hi in res})


if __name__ == "__main__":
    main()
