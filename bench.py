#!/usr/bin/env python3
"""bench.py -- 6-camera reprojection frames/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # starts its 8 ranks itself (torch.distributed.run, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W              # ... or is started as one of N ranks (WORLD_SIZE set)

Workloads (every scene: 6 pinhole cameras, 40 rendered frames at 1600x900, ~1e4 densified map
vertices from CAMA-style labels, pose rows offset from the frame stamps so every frame interpolates; camera frames
(uint8 BGR, synth.frame_pattern: the same bytes on any device, so the oracle can render them on the CPU) resident
in HBM before the timed region, mosaics stay in HBM):

  --gpus 1 (default)   BASELINE configs[1]: ONE scene.  A step = one pass of the hot path over its 40 frames.
  --gpus N > 1         BASELINE configs[2]: the FIXED 73-scene sweep (scene id = seed 0..72), sharded over the ranks by
                       shard.assign_scenes (longest-processing-time first) => "scaling": "strong".  A step = every rank
                       renders all of its scenes once.  No data-path collective.  A second, nested measurement
                       ("stress") is BASELINE configs[4]: ONE scene of 1e6 random vertices x 1000 frames, cut into
                       contiguous frame ranges (shard.frame_ranges), also fixed total work.
  --scenes S / --shard-frames / --map ...   the same machinery on other fixed workloads (any N).

One STEP of a scene = ClipManager.render_clip("cama"): frame poses for all frames (host: vectorised seek+slerp,
float32 cast, float32 inverse) -> cama_pipeline_render (bin -> overlay, one launch each per <= frames-per-call frames).
value = frames rendered by all ranks / max-over-ranks wall time of the K steps (barrier + synchronize on both sides).

Verification (untimed, BEFORE the timed region): every scene is rendered once on the plain path and hashed
(shard.overlay_hash); the per-SCENE hashes travel in the job's single all_gather (RCCL) next to the metrics and rank 0
compares them with tests/golden/scene_hashes.json -- the hashes of the ORACLE's render of the same scenes
(tests/golden/gen_scene_hashes.py, CPU) -- so a wrong shard, a scene rendered twice or not at all, or wrong pixels
cannot print a number: the run fails instead.  Between the warm-up and the timed region every output buffer is
overwritten with 0xA5 (Job.poison); after the timed region every rank hashes what the timed path itself left in its output
buffers and compares it with those verified hashes (exit 3 on a difference) -- which can therefore only pass on bytes the
timed steps wrote.  CAMA_BENCH_FAULT=skip_overlay (tests) makes the timed steps render nothing and must end in exit 3.

Extra objects in the JSON line:
  roofline      dominant kernel = k_overlay, HBM-bound.  achieved = the bytes THAT kernel moves per launch -- every
                source pixel read once, every mosaic pixel written once: 36*W*H per frame (SURVEY.md 8d's image term) x
                frames per launch -- / its mean duration measured live with hipEvents on the launch stream
                (cama_profile_*; every 8th step is timed: the launch takes the event pair as the kernel's own start /
                stop events, hipExtLaunchKernelGGL).  launch_ms_min / launch_ms_max: the fastest and slowest of those
                timed launches of the full launch size (a step of a long clip is several launches over different memory; its
                shorter last launch is left out of the spread).  peak 8000 GB/s.  `traffic` is a cross-reference to the committed
                rocprofv3 --pmc run of the same configuration (`traffic_source`), not a measurement of this run.
  roofline_project   the vertex term of SURVEY.md 8d belongs to k_frames_project, not to the overlay: bytes = 13 B (16 B
                for maps that carry a draw key) x the vertices of the 64-vertex runs that survived the block cull
                (cama_bin_stats: culled runs are never fetched) + 8 B per stamp written, / that kernel's own live
                duration.  On ~1e4-vertex maps this is 0.25 % of a frame's bytes; on 1e6-vertex maps it is the term that
                used to be charged to the overlay.
  hbm_*_whole_step   (36*W*H*F + the projection's real bytes) per step / the step's wall time.
  sustained     the same step loop run again for >= 1 s of wall time (the driver's K may be milliseconds of work); `value` is
                always the exactly-K-steps region.
  reference_default / ingest   (N = 1, default workload) nested measurements of the reference's default pipeline -- raw 1600x900
                frames undistorted + resized to 960x540 inside the overlay, golden hashes verified -- and of its ingest stage
                -- device JPEG decode of a 240-image photo-like batch, byte-equality with the host decoder asserted in the run.
  placement     what the engine's buffer placement did: candidates timed, verdict, audition_ms, audition_peak_bytes, and the
                same scene un-placed (`unplaced`).
  cpu_baseline  oracle/cama_oracle.py (numpy port of the reference, per-point circle calls into C) timed on this
                box's host cores for a bounded number of passes over the same scene; rank 0, N=1 only.  `all_cores`:
                the same loop in a process pool over independent scenes (main.py:32 has no cross-scene state), one
                scene per worker, worker count stated.
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
import cama_amd                # (first: the package's GPU_MAX_HW_QUEUES default must precede the first HIP call, cama_amd/__init__.py)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SWEEP_SCENES = 73              # BASELINE configs[2]: nuScenes v1.0-test
STRESS = dict(verts=1000000, frames=1000)       # BASELINE configs[4]
N_METRICS = 24
GOLDEN_SCENES = os.path.join(REPO, "tests", "golden", "scene_hashes.json")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=40, help="rendered frames per scene")
    ap.add_argument("--verts", type=int, default=10000, help="approximate densified vertex count")
    ap.add_argument("--height", type=int, default=900)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--scenes", type=int, default=0,
                    help="total number of distinct scenes (seed = scene id), sharded over the ranks.  Default: 1 on one "
                         "GPU (BASELINE configs[1]), 73 on several (configs[2], the v1.0-test sweep: fixed total work)")
    ap.add_argument("--shard-frames", action="store_true",
                    help="strong scaling of ONE long scene: every rank renders a contiguous range of its --frames "
                         "(shard.frame_ranges) instead of whole scenes")
    ap.add_argument("--map", choices=["lanes", "random", "site"], default="lanes",
                    help="lanes: CAMA-style densified polylines along the drive (configs[1..3]); random: --verts "
                         "uniformly random map vertices over the 600 m map in random order (configs[4] stress); "
                         "site: --verts points on 50 m polylines (1 cm spacing) spread over the whole 600 m map "
                         "(configs[3]: site-aggregated labels, ~5 %% inside the crop box at a time)")
    ap.add_argument("--sites", type=int, default=0,
                    help="with --map site and --scenes K: BASELINE configs[3] as SURVEY.md D6 defines it -- S sites, each "
                         "ONE static vertex buffer (--verts points) shared by the scenes driven on it (scene k belongs "
                         "to site k %% S and has its own pose track and calibration); scenes are placed site by site "
                         "(shard.assign_scenes(site_of=...)) and a rank uploads, sorts and indexes a site's map once")
    ap.add_argument("--raw-frames", action="store_true",
                    help="frames resident at sensor size 1600x900 and resampled (undistort+resize) to --height x "
                         "--width on the device inside every step: the reference's default 540x960 pipeline")
    ap.add_argument("--unfused-resample", action="store_true",
                    help="with --raw-frames: separate resample kernel + overlay instead of the fused raw-frame overlay")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="single stream: binning and overlay of consecutive steps do not overlap")
    ap.add_argument("--audition", type=int, default=None,
                    help="candidate allocations timed per long-lived frames / mosaic buffer (the engine's MosaicPool; default "
                         "CAMA_AUDITION or 16; 0 = plain allocations)")
    ap.add_argument("--no-scene-batch", action="store_true",
                    help="several scenes per rank: one launch chain per scene (ClipManager.render_clip) instead of one "
                         "multi-scene launch chain per step (dataset.render_clips)")
    ap.add_argument("--no-stress", action="store_true", help="N > 1: skip the nested configs[4] measurement")
    ap.add_argument("--stress-frames", type=int, default=STRESS["frames"])
    ap.add_argument("--stress-verts", type=int, default=STRESS["verts"])
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed per-scene hash pass")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (0 disables)")
    ap.add_argument("--cpu-pool-seconds", type=float, default=8.0,
                    help="budget of the all-core CPU figure (process pool over scenes; 0 disables)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="workers of the all-core CPU figure (0 = every core "
                    "that fits the host memory)")
    ap.add_argument("--segments", action="store_true",
                    help="EXTENSION (no reference semantics, SURVEY.md D1): also join neighbouring points of a polyline by "
                         "one-pixel Bresenham segments (CAMA_BIN_SEGMENTS) -- BASELINE.json's north_star wording; the bytes are "
                         "checked against the oracle's own restatement in the test-suite, not against golden hashes")
    ap.add_argument("--wu", action="store_true",
                    help="with --segments: the anti-aliased variant (Wu lines blended once by coverage, CAMA_BIN_SEGMENTS_WU)")
    ap.add_argument("--plan", action="store_true",
                    help="no GPU needed: print, as one JSON object, what every rank of `--gpus N` would hold -- its scenes, "
                         "resident frame / mosaic / map bytes, stamp scratch (worst case and, for planned site-sized maps, "
                         "the measured demand) -- against the HBM of one MI355X, and exit 0 (1 if a rank does not fit)")
    ap.add_argument("--hbm-gb", type=float, default=288.0, help="--plan: HBM per GPU to plan against")
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 default workload: skip the nested `reference_default` (raw 1600x900 -> 960x540 pipeline) and "
                         "`ingest` (device JPEG decode) measurements")
    ap.add_argument("--sustain-seconds", type=float, default=1.0,
                    help="after the K timed steps, run the same loop for at least this long and report it as "
                         "`sustained` (0 disables)")
    return ap.parse_args(argv)


def workload_key(frames, verts, width, height, map_kind, raw=False, unit="scene"):
    """Name of a workload in the golden hash files: everything the rendered bytes depend on, plus what one hashed unit
    is -- a whole scene (all its frames' mosaics) or one frame position of a frame-sharded scene."""
    return (f"map={map_kind},verts={verts},frames={frames},{width}x{height}" + (",raw1600x900" if raw else "") +
            (",per-frame" if unit == "frame" else ""))


def _segments(args):
    """False, True (Bresenham) or "wu" (anti-aliased): what ClipManager.render_clip(segments=...) takes."""
    if getattr(args, "wu", False):
        return "wu"
    return bool(getattr(args, "segments", False))


def site_of_scene(args, seed):
    """Site id of scene `seed` (--sites S > 0: scene k drives on site k % S), else None."""
    S = getattr(args, "sites", 0)
    return seed % S if S > 0 else None


def args_key(args, unit="scene"):
    """workload_key of an argument set (+ the site count, which changes which map a scene is rendered on)."""
    key = workload_key(args.frames, args.verts, args.width, args.height, args.map, raw=getattr(args, "raw_frames", False),
                       unit=unit)
    return key + (f",sites={args.sites}" if getattr(args, "sites", 0) > 0 else "") + ({False: "", True: ",segments", "wu": ",segments=wu"}[_segments(args)])


def replace_map(cm, args, seed):
    """--map random / site: replace the clip's static map through the reference's public per-clip dict
    (cama/dataset.py:13-24).  Seeded by the scene id; numpy only, so the oracle side builds the same map."""
    site = site_of_scene(args, seed)
    if site is not None:
        seed = site                                                # every scene of a site carries the SITE's labels
    if args.map == "site":
        # site-aggregated map: long polylines with random headings all over the 600 m extent, stored polyline-major
        rng = np.random.default_rng(2000 + seed)
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
        # stress map: uniformly random vertices, no spatial coherence between consecutive draw indices
        rng = np.random.default_rng(1000 + seed)
        pts = np.stack([rng.uniform(-300, 300, args.verts), rng.uniform(-300, 300, args.verts),
                        rng.normal(0, 0.05, args.verts)], axis=-1).astype(np.float32)
        half = args.verts // 2
        cm.instance_maps["cama"] = [{"class": "lane_marking", "points": pts[:half]},
                                    {"class": "Road_teeth", "points": pts[half:]}]


def write_scene_clip(args, seed, clip):
    """The synthetic clip directory of scene `seed` (no GPU, no torch): shared by build_scene and the golden scripts."""
    from cama_amd.synth import make_clip
    n_lines = max(2, round(args.verts / 500)) if args.map == "lanes" else 4
    kw = {}
    if site_of_scene(args, seed) is not None:
        # scenes of one site: each its own drive (start point seeded by the scene id) somewhere on the site
        r = np.random.default_rng(7000 + seed)
        kw["world_anchor"] = (float(r.uniform(-250.0, 100.0)), float(r.uniform(-250.0, 100.0)))
    # CAMA labels are densified at 0.1 BEV px = 1 cm: a 5 m polyline of 11 vertices gives ~500 points
    make_clip(clip, n_frames=args.frames + 1, seed=seed, n_lines=n_lines, verts_per_line=11, line_len_m=5.0,
              raster_size=3000, origin_size=(900, 1600), with_nuscenes=False, extra_labels=False, **kw)


def build_scene(args, seed, device, frame_range=None):
    """ClipManager + HBM-resident frames of scene `seed`.  frame_range = (lo, hi): only the frames of rendered
    positions lo..hi-1 (image indices lo+1..hi) are made resident (frame-sharded scenes)."""
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import DeviceFrameSource
    from cama_amd.synth import DEFAULT_CAMA_CONFIGS, frame_pattern
    H, W = args.height, args.width
    tmp = tempfile.mkdtemp(prefix=f"cama_bench_s{seed}_")
    clip = os.path.join(tmp, "clip")
    write_scene_clip(args, seed, clip)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W)), clip)
    replace_map(cm, args, seed)
    raw = getattr(args, "raw_frames", False)
    fh, fw = (900, 1600) if raw else (H, W)
    per_frame = 6 * fh * fw * 3
    first_idx, n_idx = (0, args.frames + 1) if frame_range is None else (frame_range[0] + 1, frame_range[1] - frame_range[0])
    # image index i of scene `seed` = bytes [i * per_frame, (i + 1) * per_frame) of pattern stream `seed`
    frames = frame_pattern(seed, (n_idx, 6, fh, fw, 3), device, first=first_idx * per_frame)
    if raw:
        from cama_amd.frames import RawDeviceFrameSource
        assert frame_range is None
        cm.set_frame_source(RawDeviceFrameSource(frames, cm.cm_list, fused=not getattr(args, "unfused_resample", False)))
    else:
        cm.set_frame_source(DeviceFrameSource(frames, index_offset=first_idx))
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
    first = getattr(cm.frame_source(), "index_offset", 0)      # frame-sharded scenes hold image indices first.. only
    n_host = min(int(frames.shape[0]), 48)                      # a bounded sample: at most 48 frames cross to the host
    host_frames = frames[:n_host].cpu().numpy()
    stamps, poses = O.pose_track(clip, att, cm.configs, "cama")
    secs = O.sensor_seconds(att, cm.configs["camera_main"], sync=True)
    done, t0, t_geom = 0, time.perf_counter(), 0.0
    while True:
        for idx in range(max(1, first), min(len(secs), first + n_host)):
            g0 = time.perf_counter()
            w2c = O.frame_world2chassis(stamps, poses, secs[idx])
            cropped = O.crop_instances(O.transform_instances(static, w2c))
            maps_2d = O.project_all(cropped, cams)
            t_geom += time.perf_counter() - g0
            imgs = {}
            for c, cam in enumerate(cams):
                img = host_frames[idx - first, c].copy()        # stands in for imread+remap (frames are pre-decoded)
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


def _cpu_pool_worker(job):
    """One worker of the all-core CPU figure: its own scene (seed = worker id), a few resident frames, the reference-
    structured loop of cpu_baseline until the budget is spent.  No torch, no GPU: numpy + oracle only."""
    argd, seed, n_resident, budget_s = job
    import argparse
    import tempfile as _tf
    from oracle import cama_oracle as O
    from cama_amd.synth import CAMERA_NAMES, DEFAULT_CAMA_CONFIGS, frame_pattern_np
    args = argparse.Namespace(**argd)
    H, W = args.height, args.width
    clip = os.path.join(_tf.mkdtemp(prefix=f"cama_cpu_s{seed}_"), "clip")
    write_scene_clip(args, seed, clip)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
    labels = json.load(open(os.path.join(clip, "maps", "map_labels.json")))
    bev = np.load(os.path.join(clip, "maps", "vision_road_mlp_ft.npy"))
    static = O.static_map_cama(bev, labels)
    stamps, poses = O.pose_track(clip, att, dict(DEFAULT_CAMA_CONFIGS), "cama")
    secs = O.sensor_seconds(att, "camera_front", sync=True)
    per_frame = 6 * H * W * 3
    host = frame_pattern_np(seed, (n_resident, 6, H, W, 3), first=per_frame)          # image indices 1..n_resident
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        idx = 1 + done % (len(secs) - 1)
        w2c = O.frame_world2chassis(stamps, poses, secs[idx])
        maps_2d = O.project_all(O.crop_instances(O.transform_instances(static, w2c)), cams)
        imgs = {}
        for c, cam in enumerate(cams):
            img = host[(idx - 1) % n_resident, c].copy()
            imgs[cam["name"]] = O.render_instances(img, maps_2d[cam["name"]])
        O.mosaic(imgs)
        done += 1
    return done, time.perf_counter() - t0


def usable_cores():
    """Cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (cpu.max = "quota
    period"; the GPU boxes show 256 cores and a quota of 16: 64 / 128 / 256 workers measured 685 / 640 / 594 frames/s
    against 1052 with 16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(-(-int(txt[0]) // int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, -(-q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_all_cores(args, budget_s):
    """SURVEY.md 8d(c) / BASELINE.md 3: process pool over independent scenes, one per worker, every worker the
    single-thread loop of cpu_baseline.  Workers = cores, capped so that their resident frames fit a quarter of the
    free host memory.  Returns the aggregate frames/s (sum of frames / the slowest worker's time)."""
    import multiprocessing as mp
    n_resident = 2
    per_worker = n_resident * 6 * args.height * args.width * 3 * 3 + (400 << 20)      # frames + copies + interpreter
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 64 << 30
    cores = os.cpu_count() or 1
    usable = usable_cores()
    workers = args.cpu_workers or max(1, min(usable, int(avail // 4 // per_worker)))
    argd = {k: getattr(args, k) for k in ("frames", "verts", "height", "width", "map")}
    env_keep = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in env_keep:                                  # one thread per worker: the pool is the parallelism
        os.environ[k] = "1"
    try:
        ctx = mp.get_context("spawn")                   # the parent holds a GPU context: never fork it
        t0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            res = pool.map(_cpu_pool_worker, [(argd, w, n_resident, budget_s) for w in range(workers)], chunksize=1)
        wall = time.perf_counter() - t0
    finally:
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    frames = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {"value": frames / slowest, "unit": "frames/s", "cores": workers, "host_cores": cores, "usable_cores": usable,
            "kind": "port",
            "sample": f"{workers} worker processes (spawn; usable cores = affinity capped by the cgroup quota = {usable} of "
                      f"{cores}), one independent scene each, {frames} frames in {slowest:.1f} s of "
                      f"timed loop ({wall:.1f} s incl. start-up and scene generation); same per-frame loop as the 1-core figure"}


def pmc_traffic(config_key):
    """(HBM bytes per overlay launch, source) from the newest committed rocprofv3 --pmc run (profiles/r*_pmc_traffic.json)
    of the same configuration, else (None, None).  A cross-reference, not a measurement of the current run."""
    import glob
    for p in sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9]*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(p))
            if rec.get("config") == config_key:
                return rec["bytes_per_launch"], f"profiles/{os.path.basename(p)} (" + rec.get("source", "rocprofv3 --pmc") + ")"
        except (OSError, ValueError, KeyError):
            pass
    return None, None


class Job:
    """One timed workload on this rank: a list of (scene id, ClipManager, frames) and how to step through them."""

    def __init__(self, args, scene_ids, device, frame_range=None):
        import torch
        from cama_amd import runtime
        self.args, self.device, self.frame_range = args, device, frame_range
        self.scenes = [(sid,) + build_scene(args, sid, device, frame_range) for sid in scene_ids]
        self.eng = runtime.engine()
        self.pipelined = not args.no_pipeline and not (args.raw_frames and args.unfused_resample)
        self.N = 0
        self.poses = {}
        for sid, cm, _, _ in self.scenes:
            idx, w2c = cm.frame_poses("cama")
            assert len(idx) == args.frames, (len(idx), args.frames)
            self.N = max(self.N, cm._static("cama").device().N)
        lo, hi = frame_range if frame_range is not None else (0, args.frames)
        self.lo, self.hi, self.F = lo, hi, hi - lo
        self.out = None
        self.outs = None
        self.group_frames = None                                        # (frames per multi-scene launch: the library's 16 384)
        self.batched = False
        # the caller's own (unplaced) frames of scene 0, kept for the "unplaced" comparison leg; the frame source may read a
        # placed copy from the first render on (DeviceFrameSource.place_for)
        self.orig_frames = self.scenes[0][2] if self.scenes else None

    def allocate(self):
        """The job's output buffers (and with them the engine's placement auditions: transient candidate allocations) -- apart
        from __init__ so that main() can let the ranks of a node do this a few at a time."""
        args, frame_range = self.args, self.frame_range
        if self.scenes and self.F:
            # several whole scenes per rank: ONE multi-scene launch chain per step, every scene into its own mosaic
            site_like = args.map in ("site", "random") and args.verts >= 65536     # (render_clips would decline: per-scene launches)
            multi = (len(self.scenes) > 1 and frame_range is None and not getattr(args, "no_scene_batch", False)
                     and not getattr(args, "raw_frames", False) and not _segments(args) and not site_like)
            if multi:
                # the product's own buffers: views of the engine's pooled, placed mosaics (cama_amd.dataset.clip_mosaics)
                from cama_amd.dataset import clip_mosaics
                self.outs = clip_mosaics([cm for _, cm, _, _ in self.scenes], "cama")
                self.out = self.outs[0]
                self.batched = self.step_batched()
                self.eng.join()
                if not self.batched:
                    self.outs = None
            else:
                self.out = self._first_render(0)                          # shared by the scenes of per-scene launches

    def _poses(self, cm):
        if self.frame_range is None:
            return None
        idx_all, w2c_all = cm.frame_poses("cama")                       # this rank's slice of the clip's poses
        hit = self.poses.get(id(cm))
        if hit is None or hit[0] is not w2c_all:                        # (the same slice objects while the clip's poses are the same)
            hit = self.poses[id(cm)] = (w2c_all, (idx_all[self.lo:self.hi], w2c_all[self.lo:self.hi]))
        return hit[1]

    def _first_render(self, k):
        """Scene k's mosaic buffer the way any caller of the public surface gets one: ClipManager.render_clip(out=None) returns
        a view of one of the ENGINE's pooled, placed buffers (cama_amd.engine.MosaicPool: the fastest of CAMA_AUDITION candidate
        allocations for this source, chosen once per shape per process; the resident frames are placed against it the same way,
        DeviceFrameSource.place_for).  The bench keeps that buffer and passes it back as `out=` on every step."""
        sid, cm, _, _ = self.scenes[k]
        _, out = cm.render_clip("cama", pipelined=False, poses=self._poses(cm), segments=_segments(self.args))
        return out

    def step_batched(self):
        from cama_amd.dataset import render_clips
        return render_clips([cm for _, cm, _, _ in self.scenes], "cama", self.outs, pipelined=self.pipelined,
                            max_frames_per_launch=self.group_frames)

    def step(self, out=None):
        if self.batched and out is None:
            self.step_batched()
            return
        for k, (sid, cm, _, _) in enumerate(self.scenes):
            if not self.F:
                continue
            dst = self.out if out is None else out
            cm.render_clip("cama", out=dst, pipelined=self.pipelined, poses=self._poses(cm), segments=_segments(self.args))

    def run(self, steps, warmup, sync_all, prof_every):
        import ctypes
        from cama_amd import _lib
        L = _lib.lib()
        # The interpreter's cyclic collector is parked for warm-up + timed region, as timeit does: a full collection over the
        # objects torch and numpy bring along takes ~10 ms, longer than the 20 timed steps of the headline together (one run in
        # 40 reported half the frames/s of its neighbours with an unchanged kernel time: profiles/r04_process_distribution.txt).
        # Collected BEFORE the warm-up, not between warm-up and timed region: with a pause of tens of milliseconds there the timed
        # steps ran slower at an unchanged kernel time (120-122 k -> 112-114 k frames/s, twelve of twelve processes).
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._run_timed(L, steps, warmup, sync_all, prof_every)
        finally:
            if gc_was_on:
                gc.enable()

    def _run_timed(self, L, steps, warmup, sync_all, prof_every):
        import ctypes
        for _ in range(warmup):
            self.step()
        self.eng.join()
        # Every byte of every output buffer is overwritten with a pattern no render produces BEFORE the timed region: the
        # hash check after it (timed_output_hashes) can then only pass on bytes the timed steps themselves wrote -- the
        # verified render of scene_hashes() and the warm-up's output are gone.
        self.poison()
        step = self.step
        if os.environ.get("CAMA_BENCH_FAULT") == "skip_overlay":       # fault injection (tests): the timed steps render nothing
            step = lambda: None
        verbose = os.environ.get("CAMA_BENCH_STEP_TRACE") == "1"                   # diagnostic: host time of every step's issue
        trace = []
        sync_all()
        t0 = time.perf_counter()
        for k in range(steps):
            if prof_every > 0:                                          # live hipEvent timing of every n-th overlay
                L.cama_profile_enable(1 if k % prof_every == 0 else 0)
            step()
            trace.append(time.perf_counter())                           # (~0.1 us: how long the HOST took to issue the step)
        self.eng.join()
        trace.append(time.perf_counter())
        sync_all()
        dt = time.perf_counter() - t0
        issue = np.diff(np.asarray([t0] + trace[:-1])) * 1e6 if steps else np.zeros(0)
        self.host_issue_us = {"mean": float(issue.mean()) if issue.size else 0.0, "max": float(issue.max()) if issue.size else 0.0,
                              "median": float(np.median(issue)) if issue.size else 0.0}
        if verbose:
            ts = [t0] + trace + [t0 + dt]
            print("step trace (us): issue of each step, join, final sync: " +
                  " ".join("%.0f" % ((b - a) * 1e6) for a, b in zip(ts, ts[1:])), file=sys.stderr, flush=True)
        cap = 8192
        each = (ctypes.c_double * cap)()
        ov_n = ctypes.c_int32(0)
        pj_ms, pj_n = ctypes.c_double(0.0), ctypes.c_int32(0)
        L.cama_profile_collect_each(each, cap, ctypes.byref(ov_n))
        L.cama_profile_collect_project(ctypes.byref(pj_ms), ctypes.byref(pj_n))
        L.cama_profile_enable(0)
        self.project_ms, self.project_n = pj_ms.value, pj_n.value
        got = [each[k] for k in range(min(cap, ov_n.value))]
        # per-launch spread of the timed overlay launches (a step of a long clip is several launches over different memory)
        # -- over launches of the FULL launch size: a long clip's last launch is shorter (1000 frames = 7 x 128 + 104) and
        # would otherwise be the minimum
        full = got
        bounds = getattr(self.out, "bounds", None)
        if bounds and len(bounds) > 2 and len(got) % (len(bounds) - 1) == 0:
            sizes = [b - a for a, b in zip(bounds, bounds[1:])]
            full = [t for k, t in enumerate(got) if sizes[k % len(sizes)] == max(sizes)] or got
        self.overlay_each = {"min": min(full), "max": max(full), "n": len(got)} if got else {"min": 0.0, "max": 0.0, "n": 0}
        return dt, float(sum(got)), len(got)

    def poison(self):
        import torch
        for t in [self.out] + list(self.outs or []):
            if t is not None:
                t.fill_(0xA5)
        torch.cuda.synchronize(self.device)

    def sustain(self, seconds, steps_hint, dt_hint, sync_all):
        """The same loop, un-profiled, for about `seconds` of wall time: (steps, seconds).  The number of steps is fixed
        BEFORE the region from the K timed steps' rate, and the region is bracketed like theirs (barrier + synchronize on both
        sides), so it is a timed region of exactly that many steps -- the one `value` is taken from when the driver's K steps
        are too short to carry a number (main(): < 50 ms)."""
        import gc
        if seconds <= 0 or not self.scenes or not self.F:
            for _ in range(4):                  # (a rank without work still pairs up with the others' two regions)
                sync_all()
            return 0, 0.0
        per = max(1e-6, dt_hint / max(1, steps_hint))
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()

        def region(n):
            sync_all()
            t0 = time.perf_counter()
            for _ in range(n):
                self.step()
            self.eng.join()
            sync_all()
            return time.perf_counter() - t0
        try:
            # a short calibration region first (the K steps carried profiling events and pipeline fill: their rate undershoots),
            # then THE region; every rank runs both, so the barriers pair up whatever the ranks' rates are
            n0 = max(1, int(0.2 * seconds / per + 0.5))
            d0 = region(n0)
            n = max(1, int(1.15 * seconds / max(1e-6, d0 / n0) + 1.0))
            return n, region(n)
        finally:
            if gc_was_on:
                gc.enable()

    def unplaced_leg(self, steps, warmup, sync_all, prof_every):
        """What a caller gets WITHOUT the engine's placement (the `--audition 0` figure, printed beside the default one): the
        same scene rendered from the caller's own frames tensor into a plain torch.empty mosaic, K timed steps bracketed and
        profiled like the main region.  Single-scene, pre-resized jobs only.  Returns a dict or None."""
        import torch
        from cama_amd.frames import DeviceFrameSource
        if (len(self.scenes) != 1 or self.frame_range is not None or getattr(self.args, "raw_frames", False)
                or self.orig_frames is None or not self.F or isinstance(self.out, (list, tuple)) or hasattr(self.out, "chunks")):
            return None
        sid, cm, _, _ = self.scenes[0]
        placed_src = cm.frame_source()
        env = os.environ.get("CAMA_AUDITION")
        os.environ["CAMA_AUDITION"] = "0"
        keep_out = self.out
        keep_prof = (self.project_ms, self.project_n, self.overlay_each, self.host_issue_us)
        try:
            cm.set_frame_source(DeviceFrameSource(self.orig_frames, index_offset=getattr(placed_src, "index_offset", 0)))
            self.out = torch.empty(tuple(keep_out.shape), dtype=torch.uint8, device=self.device)
            dt, ov_ms, ov_n = self.run(steps, max(warmup, 14), sync_all, prof_every)     # (14: the mapping trials of a new pair)
            return {"seconds": dt, "steps": steps, "frames_per_s": self.F * steps / dt if dt > 0 else 0.0,
                    "overlay_ms_mean": ov_ms / max(1, ov_n), "overlay_launches_timed": ov_n}
        finally:
            self.out = keep_out
            self.project_ms, self.project_n, self.overlay_each, self.host_issue_us = keep_prof
            cm.set_frame_source(placed_src)
            if env is None:
                os.environ.pop("CAMA_AUDITION", None)
            else:
                os.environ["CAMA_AUDITION"] = env

    def unmemoised_leg(self, steps, warmup, sync_all):
        """The K steps once more with cama_amd.dataset.POSE_MEMO = False: every step recomputes its poses (seek + slerp + float32 inverse) and,
        the pose arrays being new objects, goes through ClipManager.render_clip's full path instead of the memoised launch
        list -- the round-4 step, printed beside the default one."""
        from cama_amd import dataset as cama_dataset
        cama_dataset.POSE_MEMO = False
        keep_prof = (getattr(self, "project_ms", 0.0), getattr(self, "project_n", 0), getattr(self, "overlay_each", None),
                     getattr(self, "host_issue_us", None))
        try:
            dt, _, _ = self.run(steps, warmup, sync_all, 0)
            n = len(self.scenes)
            return {"seconds": dt, "steps": steps, "frames_per_s": self.F * steps * n / dt if dt > 0 else 0.0,
                    "ms_per_step": dt / max(1, steps) * 1e3, "host_issue_us": self.host_issue_us}
        finally:
            cama_dataset.POSE_MEMO = True
            self.project_ms, self.project_n, self.overlay_each, self.host_issue_us = keep_prof

    def projection_bytes(self):
        """Untimed: one plain render of the first scene, then cama_bin_stats -> (vertex bytes read, stamp bytes written)
        per FRAME, averaged over the launch, + the raw stats dict."""
        import torch
        if not self.scenes or not self.F:
            return 0.0, 0.0, None
        sid, cm, _, _ = self.scenes[0]
        if getattr(self.args, "raw_frames", False):
            return 13.0 * self.N, 0.0, None                      # (raw path: stats are not wired; vertex term only)
        idx_all, w2c_all = cm.frame_poses("cama")
        per = max(1, min(self.F, int(self.frames_per_launch() + 0.5)))
        lo = self.lo
        while True:
            poses = (idx_all[lo:lo + per], w2c_all[lo:lo + per])
            try:
                cm.render_clip("cama", out=self.out[:per], pipelined=False, poses=poses, frames_per_launch=per,
                               segments=_segments(self.args))
                break
            except torch.OutOfMemoryError:                      # (ranks sharing one GPU: the single-stream scratch on top of
                if per == 1:                                    # the pipeline's slots may not fit; fewer frames do)
                    raise
                self.eng.shrink_frames_per_call()
                per = max(1, per // 2)
        torch.cuda.synchronize(self.device)
        st = self.eng.bin_stats()
        if st is None or not st["frames"]:
            return 13.0 * self.N, 0.0, st
        return st["vertex_bytes_read"] / st["frames"], 8.0 * st["stamps"] / st["frames"], st

    def scene_hashes(self, sample_frames=None):
        """Untimed: render every scene once more (plain single-stream path) and hash it.  Whole-scene jobs: one
        (scene id, lo, hi) per scene.  Frame-sharded jobs: one (frame position, lo, hi) per sampled frame in range."""
        import torch
        from cama_amd import shard
        out = []
        for sid, cm, _, _ in self.scenes:
            if not self.F:
                continue
            poses = None
            if self.frame_range is not None:
                idx_all, w2c_all = cm.frame_poses("cama")
                poses = (idx_all[self.lo:self.hi], w2c_all[self.lo:self.hi])
            self.out.zero_()
            cm.render_clip("cama", out=self.out, pipelined=False, poses=poses, segments=_segments(self.args))
            torch.cuda.synchronize(self.device)
            if self.frame_range is None:
                out.append((sid,) + shard.overlay_hash(self.out))
            else:
                for f in sample_frames or []:
                    if self.lo <= f < self.hi:
                        out.append((f,) + shard.overlay_hash(self.out[f - self.lo]))
        return out

    def timed_output_hashes(self, sample_frames=None):
        """Hashes of what the LAST timed step left in its output buffers (same units as scene_hashes): the timed path --
        pipelined, multi-scene chains -- must have produced the bytes the plain path produced before the timed region.
        Only scenes whose mosaic is still there are reported (several scenes sharing one buffer: the last one)."""
        import torch
        from cama_amd import shard
        torch.cuda.synchronize(self.device)
        out = []
        if not self.scenes or not self.F:
            return out
        if self.frame_range is not None:
            for f in sample_frames or []:
                if self.lo <= f < self.hi:
                    out.append((f,) + shard.overlay_hash(self.out[f - self.lo]))
            return out
        bufs = self.outs if self.batched else None
        if bufs:
            return [(sid,) + shard.overlay_hash(bufs[k]) for k, (sid, _, _, _) in enumerate(self.scenes)]
        return [(self.scenes[-1][0],) + shard.overlay_hash(self.out)]

    def frames_per_launch(self):
        if not self.scenes or not self.F:
            return 0.0
        cm = self.scenes[0][1]
        if self.batched:                                                # scenes per launch x frames per scene
            per = max(1, min(1024, (self.group_frames or 16384) // self.F))
            groups = -(-len(self.scenes) // per)
            return self.F * len(self.scenes) / float(groups)
        per_call = max(1, min(self.F, self.eng.max_frames_per_call(cm._static("cama").device(), cm._rig(),
                                                                   pipelined=self.pipelined)))
        return self.F / float(-(-self.F // per_call))                   # render_clip splits big clips into launches

    def scratch_bytes(self):
        """Device bytes of stamp scratch this rank's engine holds (the pipeline's own demand-sized buffers + the
        single-stream buffer)."""
        return self.eng.scratch_bytes()

    def free(self):
        self.scenes, self.out, self.outs = [], None, None


def reference_default_leg(device, sync_all, prof_every, steps=200, warmup=10):
    """The reference's DEFAULT pipeline (cama/reproject.py:164,232-240: frames arrive at sensor size 1600x900 and are undistorted
    + resized to output_size 540x960 every frame) as a nested measurement of the N = 1 line: the same scene machinery with
    `--raw-frames --height 540 --width 960`, the fused 3:5 raw overlay, hashes checked against the oracle's golden render
    (tests/golden/scene_hashes.json, key `...,960x540,raw1600x900`) before AND after the timed region.  Returns a dict."""
    import torch
    from cama_amd import shard
    a = parse_args([])
    a.raw_frames, a.height, a.width = True, 540, 960
    t_all = time.perf_counter()
    job = Job(a, [0], device)
    job.allocate()
    key = args_key(a)
    hashes = job.scene_hashes()
    dt, ov_ms, ov_n = job.run(steps, warmup, sync_all, prof_every)
    want = {u: (lo, hi) for u, lo, hi in hashes}
    after = job.timed_output_hashes()
    golden = shard.load_golden_hashes(GOLDEN_SCENES, key)
    check = shard.verify_hashes({u: (lo, hi) for u, lo, hi in hashes}, golden, range(1))
    check["golden"] = f"tests/golden/scene_hashes.json[{key!r}]" if golden else None
    check["timed_output_equals_verified_render"] = all(want.get(u) == (lo, hi) for u, lo, hi in after) and bool(after)
    F = job.F
    per_frame = 3 * 6 * (900 * 1600 + a.height * a.width)
    fpl = job.frames_per_launch()
    ov = ov_ms / max(1, ov_n)
    ach = per_frame * fpl / (ov * 1e-3) / 1e9 if ov > 0 else 0.0
    whole = per_frame * F * steps / dt / 1e9 if dt > 0 else 0.0
    out = {"workload": "the reference's default pipeline: raw 1600x900 frames resident in HBM, undistort + resize to 960x540 "
                       "fused into the overlay's read (k_overlay_raw35), 1 scene, 6 cams x %d frames, %d verts" % (F, job.N),
           "value": F * steps / dt if dt > 0 else 0.0, "unit": "frames/s", "steps": steps, "warmup": warmup,
           "ms_per_step": dt / max(1, steps) * 1e3, "hash_check": check,
           "roofline": {"bound": "hbm", "kernel": "k_overlay_raw35", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": ov, "launches": int(ov_n), "bytes_per_launch": per_frame * fpl,
                        "launch_ms_min": job.overlay_each["min"], "launch_ms_max": job.overlay_each["max"],
                        "bytes_per_frame": "3*C*(H0*W0 + H*W): raw frame read once, resized mosaic written once", "traffic": None},
           "hbm_frac_whole_step": whole / HBM_PEAK_GBS}
    ok = not (check["mismatched"] or check["missing"]) and check["timed_output_equals_verified_render"]
    job.free()
    del job
    torch.cuda.synchronize(device)
    out["leg_seconds"] = time.perf_counter() - t_all
    return out, ok


def ingest_leg(device, batch=240, distinct=8, reps=5):
    """Frame ingest (cama/reproject.py:242-244: cv2.imread of every camera's JPEG) as a nested measurement: the device baseline-
    JPEG decoder (cama_amd.jpeg.DeviceJpegDecoder, byte-identical to libjpeg-turbo) on a batch of `batch` photo-like 1600x900
    images -- `distinct` different images (smooth gradients + sensor-like noise, quality 90, 4:2:0), repeated; every image is decoded
    independently, the compressed bytes sit in a pinned arena as ClipFrameSource's readers leave them.  The distinct images'
    device output is compared with the host decoder's (Pillow = libjpeg-turbo) IN THE RUN.  Returns (dict, ok)."""
    import io
    import torch
    from PIL import Image
    from cama_amd.jpeg import DeviceJpegDecoder
    from cama_amd import runtime
    t_all = time.perf_counter()
    rng = np.random.default_rng(0)
    y, x = np.mgrid[0:900, 0:1600].astype(np.float32)
    base = np.stack([(x * 0.16 + 20 * np.sin(y / 30)) % 256, (y * 0.28) % 256, ((x + y) * 0.1) % 256], -1)
    blobs = []
    for _ in range(distinct):
        im = np.clip(base + 6.0 * rng.standard_normal(base.shape, dtype=np.float32), 0, 255).astype(np.uint8)
        b = io.BytesIO()
        Image.fromarray(im).save(b, format="JPEG", quality=90)
        blobs.append(b.getvalue())
    datas = [blobs[k % distinct] for k in range(batch)]
    dec = runtime.engine().jpeg_decoder() if str(runtime.engine().device) == str(device) else DeviceJpegDecoder(device)
    staged = dec.stage(datas)
    out = dec.decode(staged)                                      # warm-up: table upload, lanes, scratch
    torch.cuda.synchronize(device)
    same = True
    for k in range(distinct):                                     # (BGR like cv2.imread)
        host = np.asarray(Image.open(io.BytesIO(blobs[k])).convert("RGB"))[:, :, ::-1]
        same = same and bool(np.array_equal(out[k].cpu().numpy(), host)) and bool(torch.equal(out[k], out[k + batch - distinct]))
    t0 = time.perf_counter()
    for _ in range(reps):
        out = dec.decode(staged, out=out)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / reps
    # the product's pump keeps up to three batches in flight (cama_amd/frames.py); the same here, two and three at once
    flight = {"1": batch / dt}
    try:
        staged_k = [staged] + [dec.stage(datas) for _ in range(2)]
        outs_k = [out] + [torch.empty_like(out) for _ in range(2)]
        for k in (2, 3):
            [p.result() for p in [dec.decode_async(staged_k[i], out=outs_k[i]) for i in range(k)]]
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(reps):
                [p.result() for p in [dec.decode_async(staged_k[i], out=outs_k[i]) for i in range(k)]]
            torch.cuda.synchronize(device)
            flight[str(k)] = batch * k * reps / (time.perf_counter() - t0)
        same = same and all(bool(torch.equal(o[:distinct], out[:distinct])) for o in outs_k[1:])
        del staged_k, outs_k
    except Exception as e:                                        # (a reported extra, never a reason to lose the line)
        flight["error"] = repr(e)
    in_b = float(sum(len(d) for d in datas))
    out_b = float(out.numel())
    rec = {"workload": "device JPEG decode (baseline, 4:2:0, quality 90) of %d photo-like 1600x900 images per batch (%d distinct), "
                       "compressed bytes in a pinned host arena -> BGR frames in HBM" % (batch, distinct),
           "value": batch / dt, "unit": "images/s", "six_camera_frames_per_s": batch / dt / 6.0, "ms_per_batch": dt * 1e3,
           "reps": reps, "images_per_s_by_batches_in_flight": flight, "bytes_in_per_image": in_b / batch, "bytes_out_per_image": out_b / batch,
           "hbm_GBps": (in_b + out_b) / dt / 1e9, "hbm_frac": (in_b + out_b) / dt / 1e9 / HBM_PEAK_GBS,
           "note": "bytes = compressed stream in + decoded pixels out (the algorithmic minimum; the decoder's own scratch -- unstuffed "
                   "stream, coefficients -- is not counted); latency-bound entropy decode, not an HBM-bound kernel (DESIGN.md)",
           "byte_equal_to_host_decoder": same, "host_decoder": "Pillow %s (libjpeg-turbo)" % getattr(Image, "__version__", "?"),
           "decoder_stats": dict(dec.stats)}
    del out, staged
    rec["leg_seconds"] = time.perf_counter() - t_all
    return rec, same


def stress_sample_frames(n_frames):
    """Frame positions whose hashes are checked in the frame-sharded stress: first and last frame of every rank's range
    for 1, 2, 4 and 8 ranks."""
    from cama_amd import shard
    s = set()
    for w in (1, 2, 4, 8):
        for lo, hi in shard.frame_ranges(n_frames, w):
            if hi > lo:
                s.update((lo, hi - 1))
    return sorted(s)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves --
    re-exec this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 and a free port) and
    return its exit status.  Rank 0's JSON line goes to our stdout untouched; a failing rank makes the whole job (and
    this process) exit non-zero."""
    import subprocess
    import torch
    share = os.environ.get("CAMA_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not share and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node has {have} GPU(s) visible (set CAMA_BENCH_SHARE_GPU=1 to let "
              f"the ranks share GPUs: a functional check of the N > 1 path, not a measurement)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    if share and have < args.gpus:
        env.setdefault("CAMA_BENCH_BACKEND", "gloo")       # RCCL refuses two ranks on one device
        # ranks that share a GPU also share its hardware queue slots (32; beyond them the firmware time-slices queues and
        # everything crawls): the package's default of 8 per process is for one process per GPU
        if "set by cama_amd" in (cama_amd.hw_queue_default() or ""):
            env["GPU_MAX_HW_QUEUES"] = str(max(2, min(cama_amd.HW_QUEUES, 24 * max(have, 1) // args.gpus)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def stagger_allocate(job, rank, world, sync_all, group=4):
    """Job.allocate() for at most `group` ranks of the node at a time: the placement auditions hold transient candidate
    buffers (bounded per GPU by the engine: half of what is free; `--plan` counts them) and time launches into them -- eight
    processes doing that at once share the host's driver threads and PCIe config traffic for nothing."""
    if world <= group:
        job.allocate()
        return
    for g in range(0, world, group):
        if g <= rank < g + group:
            job.allocate()
        sync_all()


def bind_this_rank(torch, local, local_world, n_dev, share):
    """shard.bind_rank with the PCI addresses of the local ranks' GPUs (rank k drives GPU k); CAMA_BENCH_NO_AFFINITY=1 skips."""
    from cama_amd import shard
    if os.environ.get("CAMA_BENCH_NO_AFFINITY") == "1":
        return {"cpus": sorted(os.sched_getaffinity(0)), "numa_node": -1, "bound": False}
    pci = []
    for k in range(local_world):
        try:
            p = torch.cuda.get_device_properties(k % n_dev if share else k)
            pci.append("%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id))
        except Exception:
            pci.append(None)
    return shard.bind_rank(local, local_world, pci)


def plan(args):
    """`bench.py --gpus N --plan`: the memory every rank of the job needs, computed on the host (no GPU, no torch.cuda):
    which scenes it gets (the same shard.assign_scenes / frame_ranges calls main() makes), what is resident in HBM during
    the timed region, and the stamp scratch.  An 8-GPU node's first run must not die on memory."""
    from cama_amd import _lib, shard
    from cama_amd.engine import PIPELINE_DEPTH as DEPTH
    L = _lib.lib()                                    # host-side size functions of the library (no device call)
    world = args.gpus
    H, W, C = args.height, args.width, 6
    frame_b = C * H * W * 3
    n_scenes = args.scenes if args.scenes > 0 else (1 if (world == 1 or args.shard_frames) else SWEEP_SCENES)
    cap = int(args.hbm_gb * 1e9)

    def scratch(N, F, planned_demand=None):
        """(worst-case bytes of ONE slot, what the pipeline will hold for its slots: cama_amd.engine.PIPELINE_DEPTH)."""
        worst = int(L.cama_render_scratch_bytes(int(N), int(F), C, H, W, 2))
        if planned_demand is not None:
            return worst, int(DEPTH * planned_demand * F)
        return worst, DEPTH * worst

    ranks = []
    if args.shard_frames:
        parts = [[0]] * world
        ranges = shard.frame_ranges(args.frames, world)
    else:
        cost = shard.scene_cost(args.frames, args.verts, W, H)
        site_of = [site_of_scene(args, k) for k in range(n_scenes)] if args.sites > 0 else None
        parts = shard.assign_scenes([cost] * n_scenes, world, site_of=site_of, site_cost=40.0 * args.verts)
        ranges = [None] * world
    site_like = args.map in ("site", "random") and args.verts >= 65536       # planned: pipeline-owned, demand-sized scratch
    # measured demand of the planned stress map: ~5.5 MB of stamp scratch per frame and slot (profiles/r04_stress_*)
    demand_per_frame = 8e6 * (args.verts / 1e6) if site_like else None
    do_stress = world > 1 and not args.no_stress and not args.shard_frames and args.scenes == 0 and args.map == "lanes"
    for r in range(world):
        mine = parts[r]
        if ranges[r] is not None:
            lo, hi = ranges[r]
            F = hi - lo
            frames_res = F * frame_b
            mosaics = F * frame_b
            fpl = min(F, 128 if site_like else F)
            worst, held = scratch(args.verts, fpl, demand_per_frame)
        else:
            F = args.frames
            frames_res = len(mine) * (F + 1) * frame_b
            batched = len(mine) > 1 and not args.no_scene_batch and not args.raw_frames and not site_like
            mosaics = (len(mine) if batched else 1) * F * frame_b
            fpl = len(mine) * F if batched else F
            worst, held = scratch(args.verts, min(fpl, 16384), demand_per_frame)
        maps = len({site_of_scene(args, k) for k in mine} if args.sites > 0 else mine) * args.verts * (13 + 16 + 48 // 64 + 1)
        rec = {"rank": r, "scenes": list(mine), "frame_range": ranges[r], "resident_frames_bytes": frames_res,
               "mosaic_bytes": mosaics, "map_bytes": maps, "frames_per_launch": fpl,
               "scratch_worst_case_one_slot": worst, "scratch_held": held}
        total = frames_res + mosaics + maps + held
        # before the timed region: candidate allocations of one mosaic / frames buffer at a time (cama_amd.engine.MosaicPool, capped by
        # the engine at half of what is free, so it cannot be what does not fit)
        one = min(F, fpl) * frame_b if ranges[r] is not None else F * frame_b
        # (one buffer: <= 16 candidates; the mosaics of a multi-scene chain / the launches of a long clip: the buffers themselves,
        # counted above, + <= 16 spare candidates)
        n_buf = len(mine) if (ranges[r] is None and batched) else (-(-F // fpl) if ranges[r] is not None and F * frame_b > (8 << 30) else 1)
        K_aud = 16 if args.audition is None else args.audition
        # one mosaic: at most K candidates (capped at half of what is free) and then K/2 candidates of the (F + 1)-frame source; several mosaics: ONE pool of max(K/4, 2) per buffer, the kept ones counted above
        # (round 6: candidates are timed four at a time and the audition stops at the first fast one / after eight alike: these are
        # upper bounds)
        cands = (K_aud + K_aud // 2) if n_buf == 1 else K_aud
        rec["placement_transient_bytes"] = 0 if (args.audition == 0 or args.raw_frames) else cands * one
        # while the engine auditions (Job.allocate, at most 4 ranks of the node at a time): the resident frames and maps + the
        # candidates; the engine itself never takes more than half (one mosaic) / three quarters (a pool) of what is free
        rec["peak_bytes_during_placement"] = frames_res + maps + rec["placement_transient_bytes"]
        total = max(total, rec["peak_bytes_during_placement"])
        if do_stress:                                               # runs after the sweep's buffers are freed
            slo, shi = shard.frame_ranges(args.stress_frames, world)[r]
            sF = shi - slo
            sw, sh = int(L.cama_render_scratch_bytes(args.stress_verts, min(sF, 128), C, H, W, 2)), int(DEPTH * 8e6 * (args.stress_verts / 1e6) * min(sF, 128))
            stress_total = 2 * sF * frame_b + args.stress_verts * 30 + sh
            rec["stress"] = {"frame_range": [slo, shi], "resident_frames_bytes": sF * frame_b, "mosaic_bytes": sF * frame_b,
                             "scratch_worst_case_one_slot": sw, "scratch_held_planned": sh, "total_bytes": stress_total}
            total = max(total, stress_total)
        rec["total_bytes"] = total
        rec["fits"] = total <= 0.92 * cap
        ranks.append(rec)
    out = {"plan": True, "gpus": world, "hbm_bytes_per_gpu": cap, "workload": args_key(args), "scenes": n_scenes,
           "ranks": ranks, "fits": all(r["fits"] for r in ranks),
           "note": "host arithmetic only: frames / mosaics / maps resident during the timed region + the stamp scratch the "
                   "pipeline's slots hold (site-sized maps: demand-sized by the pipeline, ~8 MB per frame and 10^6 "
                   "vertices measured; others: the worst case of cama_render_scratch_bytes)"}
    print(json.dumps(out))
    return 0 if out["fits"] else 1


def main():
    args = parse_args()
    if args.plan:
        sys.exit(plan(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    # stdout carries exactly ONE line, the JSON record: everything else that writes to file descriptor 1 -- RCCL prints a
    # version banner through C stdio when its first communicator is built -- is sent to stderr from here on
    sys.stdout.flush()
    record_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # CAMA_BENCH_SHARE_GPU=1 (+ CAMA_BENCH_BACKEND=gloo): run the N > 1 code path on a box with fewer GPUs than ranks
    share = os.environ.get("CAMA_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("CAMA_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if not share and local >= n_dev:
        print(f"bench.py: rank {rank} (LOCAL_RANK {local}) has no GPU: {n_dev} visible; one process per GPU is the "
              f"contract (CAMA_BENCH_SHARE_GPU=1 to share)", file=sys.stderr, flush=True)
        sys.exit(2)
    device = torch.device(f"cuda:{local % n_dev if share else local}")
    torch.cuda.set_device(device)
    os.environ["CAMA_DEVICE"] = str(device)
    # placement of long-lived buffers is the ENGINE's business (cama_amd.engine.MosaicPool); the bench only passes its
    # --audition through as the engine's environment default.  Ranks sharing one GPU (tests): no transient candidates.
    if args.audition is not None:
        os.environ["CAMA_AUDITION"] = str(max(0, args.audition))
    if share:
        os.environ["CAMA_AUDITION"] = "0"
    # one process per GPU, each on its own cores next to its GPU (the step is host-latency sensitive: ~4 us between overlays)
    affinity = bind_this_rank(torch, local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), n_dev, share)
    use_dist = world > 1 or os.environ.get("CAMA_BENCH_FORCE_DIST") == "1"   # FORCE_DIST: 1-rank RCCL group (debug)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from cama_amd import shard

    def sync_all():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    prof_every = int(os.environ.get("CAMA_BENCH_PROFILE_EVERY", "8"))
    H, W = args.height, args.width

    # ---------------------------------------------------------------- main workload
    n_scenes = args.scenes if args.scenes > 0 else (1 if (world == 1 or args.shard_frames) else SWEEP_SCENES)
    if args.shard_frames:
        n_scenes = 1
        mine = [0]                                                       # the same scene on every rank
        frange = shard.frame_ranges(args.frames, world)[rank]
    else:
        cost = shard.scene_cost(args.frames, args.verts, W, H)
        site_of = [site_of_scene(args, k) for k in range(n_scenes)] if args.sites > 0 else None
        # a rank that takes a site pays for its map once: upload + Morton sort + block index ~ 40 B per vertex moved
        assignment = shard.assign_scenes([cost] * n_scenes, world, site_of=site_of, site_cost=40.0 * args.verts)
        mine = assignment[rank]                                          # scene ids of this rank (seed = scene id)
        frange = None
    if os.environ.get("CAMA_BENCH_BAND_ROWS"):                    # A/B: force the overlay's band height (0 = the pipeline's choice)
        from cama_amd import _lib
        _lib.check(_lib.lib().cama_set_option(b"band_rows", int(os.environ["CAMA_BENCH_BAND_ROWS"])))
    job = Job(args, mine, device, frange)
    stagger_allocate(job, rank, world, sync_all)
    key = args_key(args, unit="frame" if args.shard_frames else "scene")
    samples = stress_sample_frames(args.frames) if args.shard_frames else None
    # verification first (untimed): every scene rendered once on the plain single-stream path and hashed -- what the timed
    # region then repeats is known to be right before it is timed
    hashes = [] if args.no_verify else job.scene_hashes(samples)
    dt, ov_ms, ov_n = job.run(args.steps, args.warmup, sync_all, prof_every)
    N, F = job.N, job.F
    if not args.no_verify and args.steps > 0:
        # the bytes the timed path wrote must be the bytes that were verified before it (plain path): compared on this rank
        want = {u: (lo, hi) for u, lo, hi in hashes}
        bad = [u for u, lo, hi in job.timed_output_hashes(samples) if want.get(u) != (lo, hi)]
        if bad:
            print(f"bench.py: rank {rank}: the timed path's output differs from the verified render for units {bad}",
                  file=sys.stderr, flush=True)
            sys.exit(3)
    sus_steps, sus_dt = job.sustain(args.sustain_seconds, args.steps, dt, sync_all)
    unmemoised = job.unmemoised_leg(args.steps, args.warmup, sync_all) if (world == 1 and args.steps > 0) else None
    unplaced = job.unplaced_leg(args.steps, args.warmup, sync_all, prof_every) \
        if (world == 1 and args.steps > 0 and os.environ.get("CAMA_AUDITION", "16") != "0") else None
    vbytes, sbytes, bin_stats = job.projection_bytes()
    metrics = [float(F * args.steps * len(job.scenes)), dt, ov_ms, float(ov_n), float(N),
               float(args.steps) * len(job.scenes) * shard.scene_cost(F, N, W, H), job.frames_per_launch(), 0.0,
               job.project_ms, float(job.project_n), vbytes, sbytes,
               float(F * sus_steps * len(job.scenes)), sus_dt,
               float(getattr(job.eng, "map_cache_stats", {}).get("uploads", 0)),
               float(getattr(job.eng, "map_cache_stats", {}).get("hits", 0)),
               job.overlay_each["min"], job.overlay_each["max"], float(job.scratch_bytes()),
               float(len(affinity["cpus"])), float(min(affinity["cpus"] or [-1])), float(max(affinity["cpus"] or [-1])),
               float(affinity["numa_node"]), float(bool(affinity["bound"]))]
    cm0, frames0, clip0 = (job.scenes[0][1:] if job.scenes else (None, None, None))
    slots = max(16, -(-n_scenes // world) + 1)
    report = [shard.pack_report(metrics, hashes, slots)]

    # ---------------------------------------------------------------- nested stress (configs[4]), N > 1 default only
    do_stress = world > 1 and not args.no_stress and not args.shard_frames and args.scenes == 0 and args.map == "lanes"
    if do_stress:
        job.free()
        cm0 = frames0 = None
        torch.cuda.empty_cache()
        sargs = parse_args([])
        sargs.map, sargs.verts, sargs.frames = "random", args.stress_verts, args.stress_frames
        sargs.height, sargs.width, sargs.no_pipeline = H, W, args.no_pipeline
        s_range = shard.frame_ranges(sargs.frames, world)[rank]
        sjob = Job(sargs, [0], device, s_range)
        stagger_allocate(sjob, rank, world, sync_all)
        s_steps, s_warm = max(1, args.steps // 4), max(1, args.warmup // 4)
        s_samples = stress_sample_frames(sargs.frames)
        s_hashes = [] if args.no_verify else sjob.scene_hashes(s_samples)
        sdt, sov_ms, sov_n = sjob.run(s_steps, s_warm, sync_all, prof_every)
        if not args.no_verify:
            want = {u: (lo, hi) for u, lo, hi in s_hashes}
            bad = [u for u, lo, hi in sjob.timed_output_hashes(s_samples) if want.get(u) != (lo, hi)]
            if bad:
                print(f"bench.py: rank {rank}: stress: timed output differs from the verified render for frames {bad}",
                      file=sys.stderr, flush=True)
                sys.exit(3)
        s_vb, s_sb, _ = sjob.projection_bytes()
        s_metrics = [float(sjob.F * s_steps), sdt, sov_ms, float(sov_n), float(sjob.N),
                     float(s_steps) * shard.scene_cost(sjob.F, sjob.N, W, H), sjob.frames_per_launch(), float(s_steps),
                     sjob.project_ms, float(sjob.project_n), s_vb, s_sb, 0.0, 0.0, 0.0, 0.0,
                     sjob.overlay_each["min"], sjob.overlay_each["max"], float(sjob.scratch_bytes()), 0.0, 0.0, 0.0, 0.0, 0.0]
        report.append(shard.pack_report(s_metrics, s_hashes, len(s_samples)))
        s_key = workload_key(sargs.frames, sargs.verts, W, H, "random", unit="frame")

    # ---------------------------------------------------------------- the one collective: all_gather over RCCL/xGMI
    sizes = [len(r) for r in report]
    allrep = shard.gather_reports(np.concatenate(report), device=device if backend == "nccl" else None)
    rccl_world = dist.get_world_size() if use_dist else 1

    if rank == 0:
        failures = []

        def finish(block, n_metrics, golden_key, expect_units, what):
            m, found, owner = shard.unpack_reports(block, n_metrics)
            agg = shard.reduce_metrics(m)
            golden = shard.load_golden_hashes(GOLDEN_SCENES, golden_key)
            check = None
            if not args.no_verify:
                check = shard.verify_hashes(found, golden, expect_units)
                check["golden"] = f"tests/golden/scene_hashes.json[{golden_key!r}]" if golden else None
                check["units"] = what
                if check["mismatched"] or check["missing"]:
                    failures.append(f"{what} hash check failed for {golden_key}: {check}")
            return m, agg, found, check

        m, agg, found, check = finish(allrep[:, :sizes[0]], N_METRICS, key,
                                      samples if args.shard_frames else range(n_scenes),
                                      "frame positions" if args.shard_frames else "scenes")
        fps = agg["frames_per_s"]

        def rooflines(mm, image_bytes_per_frame, overlay_kernel):
            """(roofline of the overlay, roofline of the projection, whole-step bytes per frame) from rank 0's row of
            metrics `mm`: every kernel is charged the bytes IT moves."""
            launches = max(1.0, float(mm[0, 3]))
            ov_ms_ = float(mm[0, 2]) / launches
            fpl_ = float(mm[0, 6])
            ov_bytes = image_bytes_per_frame * fpl_
            ach = ov_bytes / (ov_ms_ * 1e-3) / 1e9 if ov_ms_ > 0 else 0.0
            r_ov = {"bound": "hbm", "kernel": overlay_kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": None, "avg_launch_ms": ov_ms_,
                    "launches": int(launches), "bytes_per_launch": ov_bytes,
                    "launch_ms_min": float(mm[0, 16]), "launch_ms_max": float(mm[0, 17]),
                    "launch_max_over_min": (float(mm[0, 17]) / float(mm[0, 16])) if float(mm[0, 16]) > 0 else None,
                    "bytes_per_frame": "36*W*H = every source pixel read once + every mosaic pixel written once"}
            pj_n = float(mm[0, 9])
            pj_ms_ = float(mm[0, 8]) / pj_n if pj_n > 0 else 0.0
            pj_bytes = (float(mm[0, 10]) + float(mm[0, 11])) * fpl_
            pj_ach = pj_bytes / (pj_ms_ * 1e-3) / 1e9 if pj_ms_ > 0 else 0.0
            r_pj = {"bound": "hbm", "kernel": "k_frames_project", "achieved": pj_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": pj_ach / HBM_PEAK_GBS, "avg_launch_ms": pj_ms_, "launches": int(pj_n),
                    "bytes_per_launch": pj_bytes, "vertex_bytes_read_per_frame": float(mm[0, 10]),
                    "stamp_bytes_written_per_frame": float(mm[0, 11]),
                    "vertex_read_fraction": float(mm[0, 10]) / max(1.0, 13.0 * float(mm[0, 4])),
                    "note": "13 B per vertex of the 64-vertex runs that survived the block cull (cama_bin_stats) + 8 B "
                            "per stamp; in practice fp64-issue bound, not HBM bound (DESIGN.md section 4)"}
            return r_ov, r_pj, image_bytes_per_frame + float(mm[0, 10]) + float(mm[0, 11])

        image_bytes = 36 * W * H                                # the image term, the overlay's bytes
        r_overlay, r_project, bytes_per_frame = rooflines(m, image_bytes, "k_overlay")
        r_overlay["traffic"], r_overlay["traffic_source"] = pmc_traffic(f"N={N},F={F},{W}x{H}")
        cfg_no = 4 if args.map == "random" else 3 if args.map == "site" else 2 if n_scenes > 1 else 1
        line = {
            "metric": "6-cam frames/sec (1600x900, ~10k map verts)" if (W, H) == (1600, 900)
                      else f"6-cam frames/sec ({W}x{H})",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": agg["seconds"] / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (world > 1 or args.shard_frames or n_scenes > 1) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: %d scene(s) over %d GPU(s), 6 cams x %d frames each, %d "
                                   "densified verts, %dx%d, frames resident in HBM; fixed total work"
                                   % (cfg_no, n_scenes, world, args.frames, N, W, H),
                       "scenes": n_scenes, "frames_per_scene": args.frames, "verts": N, "width": W, "height": H,
                       "map": args.map,
                       "sharding": ("contiguous frame ranges of one scene (shard.frame_ranges)" if args.shard_frames
                                    else "whole scenes, longest-processing-time first (shard.assign_scenes)")
                                   + ", no data-path collective",
                       "streams": "2 (binning of launch k+1 overlaps overlay of launch k)" if job.pipelined else "1",
                       "launches_per_step": "1 multi-scene chain for all of a rank's scenes (dataset.render_clips)"
                                            if job.batched else "1 chain per scene (ClipManager.render_clip)"},
            "rccl_world": rccl_world, "collective": "one all_gather_into_tensor of %d int64 per rank (%s)" % (
                allrep.shape[1], "RCCL" if (use_dist and backend == "nccl") else backend if use_dist else "no group"),
            "per_rank_seconds": [float(x) for x in m[:, 1]],
            "per_rank_frames": [float(x) for x in m[:, 0]],
            "hash_check": check,
            "scene_hashes": {str(k): ["%016x" % v[0], "%016x" % v[1]] for k, v in sorted(found.items())}
                            if len(found) <= 8 else f"{len(found)} units (see hash_check)",
            "hbm_GBps_whole_step": bytes_per_frame * (float(m[0, 0]) / float(m[0, 1])) / 1e9,   # rank 0's GPU
            "hbm_frac_whole_step": bytes_per_frame * (float(m[0, 0]) / float(m[0, 1])) / 1e9 / HBM_PEAK_GBS,
            "roofline": r_overlay,
            "roofline_project": r_project,
            "scratch_bytes": int(m[0, 18]),
        }
        if args.raw_frames:
            line["config"]["workload"] += "; raw 1600x900 frames resampled on device each step"
            # this mode's kernel reads the raw frames and writes the resized mosaic: 3*C*(H0*W0 + H*W) bytes per frame
            raw_image = 3 * 6 * (900 * 1600 + H * W)
            r_overlay, r_project, raw_bytes = rooflines(m, raw_image, "k_overlay_raw*")
            r_overlay["bytes_per_frame"] = "3*C*(H0*W0 + H*W): raw frame read once, resized mosaic written once"
            line["roofline"], line["roofline_project"] = r_overlay, r_project
            bytes_per_frame = raw_bytes
            line["hbm_GBps_whole_step"] = raw_bytes * (float(m[0, 0]) / float(m[0, 1])) / 1e9
            line["hbm_frac_whole_step"] = line["hbm_GBps_whole_step"] / HBM_PEAK_GBS
        sus_frames, sus_secs = float(m[:, 12].sum()), float(m[:, 13].max())
        if sus_secs > 0:
            line["sustained"] = {"value": sus_frames / sus_secs, "unit": "frames/s", "seconds": sus_secs,
                                 "frames": sus_frames, "steps": int(round(sus_frames / max(1.0, float(F * n_scenes)))),
                                 "note": "the same step loop run for about %.1f s after the K timed steps, bracketed the same way "
                                         "(barrier + synchronize on both sides; sum of frames / slowest rank)" % args.sustain_seconds}
            # `value` / `ms_per_step` are ALWAYS the exactly-K-steps region (the contract); the long region is printed beside it
            # because K steps of a few hundred microseconds carry one step's un-overlapped binning chain and the final join
            # (~2-3 % of a 7 ms region) -- compare `sustained.value` for the steady rate.
            line["sustained"]["hbm_frac_whole_step"] = (bytes_per_frame * (float(m[0, 12]) / float(m[0, 13])) / 1e9 / HBM_PEAK_GBS
                                                        if float(m[0, 13]) > 0 else None)
            line["value_source"] = "the %d timed steps (%.1f ms)" % (args.steps, agg["seconds"] * 1e3)
        line["config"]["step"] = ("ClipManager.render_clip(out=<the engine's pooled mosaic>, pipelined): frame poses memoised per "
                                  "(track, stamps); the clip's launches worked out once and replayed, one library call each "
                                  "(cama_pipeline_render_clip)")
        if unmemoised is not None:
            line["without_memo"] = dict(unmemoised, note="the same K steps with cama_amd.dataset.POSE_MEMO = False: seek + slerp + float32 inverse "
                                                         "recomputed and the launch arguments re-derived in Python on every step "
                                                         "(what every step did until round 4)")
        line["host_issue_us"] = dict(job.host_issue_us, note="rank 0: wall time of one step()'s issue on the host inside the K timed "
                                     "steps (the GPU is paced by the host whenever this approaches the step time)")
        line["hw_queues"] = cama_amd.hw_queue_default()
        line["rank_affinity"] = {"cpus_per_rank": [int(x) for x in m[:, 19]], "first_cpu": [int(x) for x in m[:, 20]],
                                 "last_cpu": [int(x) for x in m[:, 21]], "gpu_numa_node": [int(x) for x in m[:, 22]],
                                 "bound": [bool(x) for x in m[:, 23]],
                                 "note": "every rank pins itself to a contiguous share of the cores of its GPU's NUMA node "
                                         "(cama_amd.shard.bind_rank; an even split of the allowed cores when sysfs has no NUMA "
                                         "picture); disjoint by construction"}
        if bin_stats is not None:
            line["projection_stats"] = bin_stats
        pinfo = job.eng.pipeline_info() if hasattr(job, "eng") else None
        if pinfo:
            line["band_rows"] = {"last_launch": pinfo["last_band_rows"], "tall_band_launches": pinfo["tall_band_launches"],
                                 "launches": pinfo["launches"],
                                 "note": "rows per band of the overlay, chosen per launch by the pipeline: 8 instead of 4 once the same map's "
                                         "earlier launches have shown >= 0.045 band entries per destination pixel (dense maps)"}
        try:
            periodic, first8, per_xcd = job.eng.xcd_map()
            line["xcd_dispatch"] = {"round_robin_over_8_xcds": periodic, "xcd_of_blocks_0_to_7": first8,
                                    "blocks_per_xcd": per_xcd,
                                    "note": "the overlay's XCD-contiguous mapping assumes consecutive blocks go round-robin "
                                            "over the 8 XCDs (speed only; HW_REG_XCC_ID probe)"}
        except Exception as e:
            line["xcd_dispatch"] = {"error": repr(e)}
        try:
            line["overlay_mapping"] = dict(job.eng.overlay_mapping(),
                                           note="big launches (>= 1.75 GiB): the library times the XCD-contiguous order (31) "
                                                "against round-robin chunks of 32 bands (5) on the first launches over each "
                                                "(frames, mosaic) buffer pair -- three each, medians -- and keeps the faster for "
                                                "that pair (the contiguous order's speed is a property of the buffers' physical "
                                                "placement: profiles/r04_overlay_modes.txt); reported: the pair of the most recent "
                                                "big launch; -1 = not decided (no big launch, fewer than six launches over the "
                                                "pair, or forced)")
        except Exception as e:
            line["overlay_mapping"] = {"error": repr(e)}
        log = getattr(job.eng, "audition_log", None)
        pool = getattr(job.eng, "pool", None)
        line["placement"] = {"source": "engine default" if os.environ.get("CAMA_AUDITION", "16") != "0" else "off (CAMA_AUDITION=0)",
                             "pool": dict(pool.stats, bases=len(pool.bases), bytes=pool.nbytes()) if pool is not None else None,
                             "note": "rank 0.  bench.py allocates no mosaic itself: render_clip(out=None) / clip_mosaics() hand out "
                                     "views of the engine's pooled buffers (cama_amd.engine.MosaicPool), each the fastest of N "
                                     "candidate allocations for its source (stamp-free overlay launches timed into each), and "
                                     "HBM-resident frames are moved once into the fastest of M candidates for that mosaic "
                                     "(DeviceFrameSource.place_for): the overlay's bandwidth depends on the pair's physical "
                                     "placement (profiles/r04_overlay_modes.txt section 5).  plain_first_candidate_ms = the first "
                                     "candidate, what an un-auditioned allocation would have been; `unplaced` = the same scene timed "
                                     "from the caller's own frames into a plain torch.empty mosaic (the --audition 0 figure)"}
        if pool is not None:
            line["placement"]["audition_ms"] = pool.stats["audition_seconds"] * 1e3
            line["placement"]["audition_peak_bytes"] = int(pool.stats["audition_peak_bytes"])
            line["placement"]["flat_box"] = bool(pool.flat_box)
        if log:
            line["placement"]["verdict"] = next((e.get("verdict") for e in log if e["role"] == "mosaic"), None)
            mos = [e for e in log if e["role"] == "mosaic"]
            frs = [e for e in log if e["role"] == "frames"]
            line["placement"].update({
                "buffers": len(mos), "candidates_per_mosaic": mos[0]["candidates"] if mos else 0,
                "candidates_per_frames": frs[0]["candidates"] if frs else 0,
                "first_mosaic_candidates_ms": mos[0]["ms"] if mos else None,
                "first_frames_candidates_ms": frs[0]["ms"] if frs else None,
                "kept_of_pool": mos[0].get("kept") if mos else None,
                "chosen_ms_mean": float(np.mean([e["chosen_ms"] for e in (frs or mos)])),
                "plain_first_candidate_ms": float(np.mean([e["ms"][0] for e in mos])) if mos else None})
        if unplaced is not None:
            ub = image_bytes * F / (unplaced["overlay_ms_mean"] * 1e-3) / 1e9 if unplaced["overlay_ms_mean"] > 0 else 0.0
            line["placement"]["unplaced"] = dict(unplaced, kernel_GBps=ub, kernel_frac=ub / HBM_PEAK_GBS)
        if args.sites > 0:
            line["site_maps"] = {"sites": args.sites, "verts_per_site": N,
                                 "sites_per_rank": shard.sites_per_rank(assignment, site_of),
                                 "map_uploads_per_rank": [int(x) for x in m[:, 14]],
                                 "map_cache_hits_per_rank": [int(x) for x in m[:, 15]],
                                 "note": "scenes of a site share ONE device vertex buffer per rank (content-keyed "
                                         "Engine.shared_map): uploaded, Morton-sorted and indexed once"}
            line["config"]["workload"] += "; %d site(s), scene k on site k %% %d" % (args.sites, args.sites)
            line["config"]["sharding"] = "whole scenes placed site by site (shard.assign_scenes(site_of=...)), no data-path collective"
        if _segments(args):
            line["config"]["workload"] += ("; EXTENSION: discs + anti-aliased (Wu) segments between polyline neighbours"
                                           if _segments(args) == "wu" else
                                           "; EXTENSION: discs + one-pixel Bresenham segments between polyline neighbours")
            line["config"]["extension"] = ("segments: no reference semantics (the reference draws a disc per point, "
                                           "cama/reproject.py:255-256); bytes checked against the oracle's own restatement "
                                           "in tests/test_gpu_kernels.py and tests/test_gpu_fullsize.py")
        if do_stress:
            sm, sagg, sfound, scheck = finish(allrep[:, sizes[0]:], N_METRICS, s_key, s_samples, "frame positions")
            s_rov, s_rpj, s_bpf = rooflines(sm, 36 * W * H, "k_overlay")
            line["stress"] = {
                "workload": "BASELINE configs[4]: 1 scene of %d random verts x 6 cams x %d frames at %dx%d, contiguous "
                            "frame ranges over %d GPU(s)" % (int(sm[0, 4]), sargs.frames, W, H, world),
                "value": sagg["frames_per_s"], "unit": "frames/s", "steps": int(sm[0, 7]), "scaling": "strong",
                "ms_per_step": sagg["seconds"] / max(1.0, float(sm[0, 7])) * 1e3,
                "per_rank_seconds": [float(x) for x in sm[:, 1]], "per_rank_frames": [float(x) for x in sm[:, 0]],
                "hash_check": scheck,
                "roofline": s_rov, "roofline_project": s_rpj,
                "hbm_frac_whole_step": s_bpf * (float(sm[0, 0]) / float(sm[0, 1])) / 1e9 / HBM_PEAK_GBS,
            }
        if failures:
            print("\n".join(failures), file=sys.stderr, flush=True)
            if use_dist:
                dist.destroy_process_group()
            sys.exit(3)                                          # a wrong render never prints a throughput line
        default_workload = (world == 1 and n_scenes == 1 and args.map == "lanes" and not args.raw_frames and not args.shard_frames
                            and (W, H) == (1600, 900) and not _segments(args) and not args.no_extras and not args.no_verify)
        if default_workload:
            # the reference's default pipeline and its ingest stage, in the line the driver records
            line["reference_default"], ok = reference_default_leg(device, sync_all, prof_every)
            if not ok:
                failures.append("reference_default: hash check failed: %r" % (line["reference_default"]["hash_check"],))
            try:
                line["ingest"], ok = ingest_leg(device)
                if not ok:
                    failures.append("ingest: the device JPEG decoder's output differs from the host decoder's")
            except ImportError as e:                              # (no Pillow: there is nothing to encode the batch with)
                line["ingest"] = {"skipped": repr(e)}
        if failures:
            print("\n".join(failures), file=sys.stderr, flush=True)
            sys.exit(3)
        if world == 1 and args.cpu_seconds > 0 and not args.raw_frames and cm0 is not None:
            line["cpu_baseline"] = cpu_baseline(cm0, frames0, clip0, args, args.cpu_seconds)
            line["speedup_vs_cpu_baseline"] = fps / line["cpu_baseline"]["value"]
            if args.cpu_pool_seconds > 0 and args.map == "lanes":
                try:
                    line["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(args, args.cpu_pool_seconds)
                    line["speedup_vs_cpu_all_cores"] = fps / line["cpu_baseline"]["all_cores"]["value"]
                except Exception as e:                              # a reported baseline, never a reason to lose the line
                    line["cpu_baseline"]["all_cores"] = {"error": repr(e)}
        record_out.write(json.dumps(line) + "\n")
        record_out.flush()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
