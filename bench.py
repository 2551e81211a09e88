#!/usr/bin/env python3
"""bench.py -- 6-camera reprojection frames/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): one synthetic scene per GPU -- 6 pinhole cameras,
40 rendered frames at 1600x900, ~1e4 densified map vertices (CAMA-style labels: BEV pixels + height raster),
pose rows offset from the frame stamps so every frame interpolates.  Camera frames (uint8 BGR) are resident in
HBM before the timed region; the mosaic output stays in HBM.

One STEP = one pass of the hot path over the scene:
    ClipManager.render_clip("cama")  =  frame poses for all 40 frames (host: vectorised seek+slerp, float32
    cast, float32 inverse) -> cama_render_frames (count -> scan -> fill -> overlay, one launch each for all frames).
value = frames rendered by all ranks / max-over-ranks wall time of the K steps (barrier + synchronize on both sides).
Scenes are independent, so N GPUs render N scenes (weak scaling, no data-path collective); the only collective is
one all_gather of an 8-double metric record (frames, seconds, overlay time, bytes, overlay checksum) over RCCL.

Extra objects in the JSON line:
  roofline      dominant kernel = k_overlay, HBM-bound.  achieved = algorithmic bytes per launch
                (13*N + 36*W*H per frame, SURVEY.md 8d, x frames per launch) / its mean duration measured live
                with hipEvents on the launch stream (cama_profile_*; every 8th launch is timed, a timed event pair
                is two extra barrier packets).  peak 8000 GB/s.
  cpu_baseline  oracle/cama_oracle.py (numpy port of the reference, per-point circle calls into C) timed on this
                box's host cores for a bounded number of passes over the same scene; rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=40, help="rendered frames per scene")
    ap.add_argument("--verts", type=int, default=10000, help="approximate densified vertex count")
    ap.add_argument("--height", type=int, default=900)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--scenes", type=int, default=0,
                    help="total number of distinct scenes, sharded over the ranks (default: one per rank). "
                         "73 = BASELINE configs[2], the v1.0-test sweep; every step renders all scenes of the rank")
    ap.add_argument("--shard-frames", action="store_true",
                    help="strong scaling of ONE long scene: every rank renders a contiguous range of its --frames "
                         "(shard.frame_ranges) instead of a scene of its own")
    ap.add_argument("--map", choices=["lanes", "random", "site"], default="lanes",
                    help="lanes: CAMA-style densified polylines along the drive (configs[1..3]); random: --verts "
                         "uniformly random map vertices over the 600 m map in random order (configs[4] stress); "
                         "site: --verts points on 50 m polylines (1 cm spacing) spread over the whole 600 m map "
                         "(configs[3]: site-aggregated labels, ~5 %% inside the crop box at a time)")
    ap.add_argument("--raw-frames", action="store_true",
                    help="frames resident at sensor size 1600x900 and resampled (undistort+resize) to --height x "
                         "--width on the device inside every step: the reference's default 540x960 pipeline")
    ap.add_argument("--unfused-resample", action="store_true",
                    help="with --raw-frames: separate resample kernel + overlay instead of the fused raw-frame overlay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="single stream: binning and overlay of consecutive steps do not overlap")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    return ap.parse_args()


def build_scene(args, seed, device):
    import torch
    rank = seed
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import DeviceFrameSource
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip
    H, W = args.height, args.width
    tmp = tempfile.mkdtemp(prefix=f"cama_bench_r{rank}_")
    clip = os.path.join(tmp, "clip")
    n_lines = max(2, round(args.verts / 500)) if args.map == "lanes" else 4
    # CAMA labels are densified at 0.1 BEV px = 1 cm: a 5 m polyline of 11 vertices gives ~500 points
    make_clip(clip, n_frames=args.frames + 1, seed=rank, n_lines=n_lines, verts_per_line=11, line_len_m=5.0,
              raster_size=3000, origin_size=(900, 1600), with_nuscenes=False, extra_labels=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W)), clip)
    if args.map == "site":
        # site-aggregated map: long polylines with random headings all over the 600 m extent, stored polyline-major
        rng = np.random.default_rng(2000 + rank)
        per_line = 5000                                            # 50 m at 1 cm
        n_l = max(1, args.verts // per_line)
        t = (np.arange(per_line) * 0.01)[None, :]
        p0 = rng.uniform(-290, 240, (n_l, 2))
        ang = rng.uniform(0, 2 * np.pi, n_l)
        x = p0[:, 0:1] + t * np.cos(ang)[:, None]
        y = p0[:, 1:2] + t * np.sin(ang)[:, None] + 0.3 * np.sin(t * 0.2)
        z = rng.normal(0, 0.05, (n_l, per_line))
        pts = np.stack([x, y, z], axis=-1).astype(np.float32)
        cls = ["lane_marking", "Road_teeth", "Crosswalk_Line"]
        cm.instance_maps["cama"] = [{"class": cls[i % 3], "points": pts[i]} for i in range(n_l)]
    if args.map == "random":
        # stress map: uniformly random vertices, no spatial coherence between consecutive draw indices.
        # instance_maps is the reference's public per-clip dict (cama/dataset.py:13-24): replace the static map.
        rng = np.random.default_rng(1000 + rank)
        pts = np.stack([rng.uniform(-300, 300, args.verts), rng.uniform(-300, 300, args.verts),
                        rng.normal(0, 0.05, args.verts)], axis=-1).astype(np.float32)
        half = args.verts // 2
        cm.instance_maps["cama"] = [{"class": "lane_marking", "points": pts[:half]},
                                    {"class": "Road_teeth", "points": pts[half:]}]
    gen = torch.Generator(device=device)
    gen.manual_seed(rank)
    if getattr(args, "raw_frames", False):
        from cama_amd.frames import RawDeviceFrameSource
        frames = torch.randint(0, 256, (args.frames + 1, 6, 900, 1600, 3), dtype=torch.uint8, device=device,
                               generator=gen)
        cm.set_frame_source(RawDeviceFrameSource(frames, cm.cm_list, fused=not getattr(args, "unfused_resample", False)))
    else:
        frames = torch.randint(0, 256, (args.frames + 1, 6, H, W, 3), dtype=torch.uint8, device=device, generator=gen)
        cm.set_frame_source(DeviceFrameSource(frames))
    return cm, frames, clip


def cpu_baseline(cm, frames, clip, args, budget_s):
    """Reference-structured CPU path (oracle, numpy + per-point C circle) on the same scene; bounded."""
    from oracle import cama_oracle as O
    from cama_amd.synth import CAMERA_NAMES
    H, W = args.height, args.width
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
    labels = json.load(open(os.path.join(clip, "maps", "map_labels.json")))
    bev = np.load(os.path.join(clip, "maps", "vision_road_mlp_ft.npy"))
    static = O.static_map_cama(bev, labels)                     # per-clip setup, not timed (GPU side: ClipManager())
    host_frames = frames.cpu().numpy()
    stamps, poses = O.pose_track(clip, att, cm.configs, "cama")
    secs = O.sensor_seconds(att, cm.configs["camera_main"], sync=True)
    done, t0, t_geom = 0, time.perf_counter(), 0.0
    while True:
        for idx in range(1, len(secs)):
            g0 = time.perf_counter()
            w2c = O.frame_world2chassis(stamps, poses, secs[idx])
            cropped = O.crop_instances(O.transform_instances(static, w2c))
            maps_2d = O.project_all(cropped, cams)
            t_geom += time.perf_counter() - g0
            imgs = {}
            for c, cam in enumerate(cams):
                img = host_frames[idx, c].copy()                # stands in for imread+remap (frames are pre-decoded)
                imgs[cam["name"]] = O.render_instances(img, maps_2d[cam["name"]])
            O.mosaic(imgs)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{done} frames of the same scene ({W}x{H}, 6 cams) in {dt:.1f} s; oracle/cama_oracle.py "
                      f"single thread; host has {os.cpu_count()} cores; frames pre-decoded in RAM (no JPEG/remap); "
                      f"numpy geometry stages alone (pose+transform+crop+project): {done / t_geom:.0f} frames/s",
            "geometry_only_fps": done / t_geom}


def pmc_traffic(config_key):
    """HBM bytes per overlay launch from a committed rocprofv3 --pmc run (profiles/pmc_traffic.json), else None."""
    p = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(p))
        return rec["bytes_per_launch"] if rec.get("config") == config_key else None
    except (OSError, ValueError, KeyError):
        return None


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # CAMA_BENCH_SHARE_GPU=1 + CAMA_BENCH_BACKEND=gloo: debugging aid to run the N>1 code path on a 1-GPU box
    share = os.environ.get("CAMA_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("CAMA_BENCH_BACKEND", "nccl")
    device = torch.device(f"cuda:{local % torch.cuda.device_count() if share else local}")
    torch.cuda.set_device(device)
    os.environ["CAMA_DEVICE"] = str(device)
    use_dist = world > 1 or os.environ.get("CAMA_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: 1-rank RCCL group (debug)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from cama_amd import _lib, runtime, shard
    n_scenes = args.scenes if args.scenes > 0 else world
    cost = shard.scene_cost(args.frames, args.verts, args.width, args.height)
    mine = shard.assign_scenes([cost] * n_scenes, world)[rank]           # scene ids of this rank (seed = scene id)
    if args.shard_frames:
        mine = [0]                                                       # the same scene on every rank
    scenes = [build_scene(args, sid, device) for sid in mine]
    cm, frames, clip = scenes[0] if scenes else (None, None, None)
    eng = runtime.engine()
    F = args.frames
    N = 0
    out = None
    for scm, _, _ in scenes:
        idx, _ = scm.frame_poses("cama")
        assert len(idx) == args.frames, (len(idx), args.frames)
        N = max(N, scm._static("cama").device().N)
    f_lo, f_hi = shard.frame_ranges(args.frames, world)[rank] if args.shard_frames else (0, args.frames)
    F = f_hi - f_lo                                                      # frames this rank renders per scene
    if scenes:
        out = torch.empty(eng.mosaic_shape(cm._rig(), F), dtype=torch.uint8, device=device)   # shared by the scenes

    def sync_all():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    pipelined = not args.no_pipeline and not args.raw_frames
    def step():
        for scm, _, _ in scenes:
            poses = None
            if args.shard_frames:                                        # this rank's slice of the clip's poses
                idx_all, w2c_all = scm.frame_poses("cama")
                poses = (idx_all[f_lo:f_hi], w2c_all[f_lo:f_hi])
            scm.render_clip("cama", out=out, pipelined=pipelined, poses=poses)

    for _ in range(args.warmup):
        step()
    eng.join()
    sync_all()
    L = _lib.lib()
    # live hipEvent timing of the overlay kernel: sampled (every `prof_every`-th step) because a timed event pair is
    # two extra barrier packets on the launch stream
    prof_every = int(os.environ.get("CAMA_BENCH_PROFILE_EVERY", "8"))
    t0 = time.perf_counter()
    for k in range(args.steps):
        if prof_every > 0:
            L.cama_profile_enable(1 if k % prof_every == 0 else 0)
        step()
    eng.join()
    sync_all()
    dt = time.perf_counter() - t0
    import ctypes
    ov_ms, ov_n = ctypes.c_double(0.0), ctypes.c_int32(0)
    L.cama_profile_collect(ctypes.byref(ov_ms), ctypes.byref(ov_n))
    L.cama_profile_enable(0)

    H, W = args.height, args.width
    h_lo, h_hi = shard.overlay_hash(out) if out is not None else (0, 0)   # checksum of the last mosaics (untimed)
    rec = [float(F * args.steps * len(scenes)), dt, ov_ms.value, float(ov_n.value), float(N),
           float(args.steps) * len(scenes) * shard.scene_cost(F, N, W, H), float(h_lo % 2 ** 52), float(h_hi % 2 ** 52)]
    # the one collective: metric all_gather over RCCL/xGMI
    allrec = shard.gather_records(rec, device=device if backend == "nccl" else None)
    agg = shard.reduce_metrics(allrec)

    if rank == 0:
        wall = agg["seconds"]
        fps = agg["frames_per_s"]
        bytes_per_frame = 13 * N + 36 * W * H
        launches = max(1.0, float(allrec[0, 3]))
        ov_avg_ms = float(allrec[0, 2]) / launches
        per_call = max(1, min(F, eng.max_frames_per_call(cm._static("cama").device(), cm._rig())))
        frames_per_launch = F / float(-(-F // per_call)) if F else 0.0   # render_clip splits big clips into launches
        achieved = bytes_per_frame * frames_per_launch / (ov_avg_ms * 1e-3) / 1e9 if ov_avg_ms > 0 else 0.0
        cfg_key = f"N={N},F={F},{W}x{H}"
        line = {
            "metric": "6-cam frames/sec (1600x900, ~10k map verts)" if (W, H) == (1600, 900)
                      else f"6-cam frames/sec ({W}x{H})",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.shard_frames else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d scene(s) over %d GPU(s), 6 cams x %d frames, %d densified "
                                   "verts, %dx%d, frames resident in HBM" % (4 if args.map == "random" else 3 if args.map == "site" else 2 if n_scenes > world else 1,
                                                                           n_scenes, world, F, N, W, H),
                       "scenes": n_scenes,
                       "frames_per_step": F, "verts": N, "width": W, "height": H, "map": args.map,
                       "sharding": "one scene per rank, no data-path collective",
                       "streams": "2 (binning of step k+1 overlaps overlay of step k)" if pipelined else "1"},
            "overlay_hash_per_rank": agg["hash"],
            "hbm_GBps_whole_step": bytes_per_frame * (float(allrec[0, 0]) / float(allrec[0, 1])) / 1e9,   # rank 0's GPU
            "hbm_frac_whole_step": bytes_per_frame * (float(allrec[0, 0]) / float(allrec[0, 1])) / 1e9 / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm", "kernel": "k_overlay", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(cfg_key),
                         "avg_launch_ms": ov_avg_ms, "launches": int(launches),
                         "bytes_per_launch": bytes_per_frame * frames_per_launch},
        }
        if args.raw_frames:
            line["config"]["workload"] += "; raw 1600x900 frames resampled on device each step"
        if world == 1 and args.cpu_seconds > 0 and not args.raw_frames:
            line["cpu_baseline"] = cpu_baseline(cm, frames, clip, args, args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = fps / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
