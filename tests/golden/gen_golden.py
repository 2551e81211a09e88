#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference) in this
container on tiny synthetic clips.  The reference never travels to the GPU box:
only the .npz files written next to this script do.  No-op when /root/reference
is absent.

Neither `cv2` nor `ffmpeg` is installed here, so both are replaced by stub
modules *before* the reference is imported.  The cv2 stub RECORDS every
cv2.circle call; every numeric step of the reference up to that call
(densify, pose load/compose/seek/slerp, float32 inverse, transform, crop,
project, mask, truncation, colour choice, draw order) executes for real.
What stays unpinned is OpenCV's own arithmetic (circle footprint, remap,
JPEG decode) -- see DESIGN.md "parity pins".

Fixtures (all small, deterministic, seeded):
  clip_<tag>.npz   per-clip: static maps, calibration, pose tracks, per-frame
                   world2chassis/crop/projection outputs, cv2.circle stream
  pose_seek.npz    seek_by_timestamp edge cases (exact / interpolated / raise)
  mosaic.npz       VideoGenerator.concate_image layout
Usage:  python tests/golden/gen_golden.py
"""
import json
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"


def _drop_cama_modules():
    for m in [k for k in sys.modules if k == "cama" or k.startswith("cama.")]:
        del sys.modules[m]


def bind_reference_package():
    """Make `import cama.<x>` resolve to /root/reference/cama.  The reference's `cama/` has no __init__.py (a
    namespace package), so a regular package of the same name anywhere on sys.path -- this repo's drop-in `cama/`
    shim -- would win no matter the path order; bind the name explicitly instead."""
    _drop_cama_modules()
    pkg = types.ModuleType("cama")
    pkg.__path__ = [os.path.join(REFERENCE, "cama")]
    sys.modules["cama"] = pkg


def assert_from_reference(*modules):
    for m in modules:
        f = os.path.abspath(sys.modules[m.__module__].__file__ if not isinstance(m, types.ModuleType) else m.__file__)
        assert f.startswith(os.path.abspath(REFERENCE) + os.sep), f


class _Recorder:
    def __init__(self):
        self.calls = []


def install_stubs(rec):
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1
    cv2.INTER_NEAREST = 0
    cv2.CV_32FC1 = 5
    cv2.IMREAD_ANYDEPTH = 2

    def circle(img, center, radius, color, thickness):
        rec.calls.append((int(center[0]), int(center[1]), int(radius),
                          int(color[0]), int(color[1]), int(color[2]), int(thickness),
                          type(center[0]).__name__))
        return img

    def imread(path, flags=None):
        return np.zeros((900, 1600, 3), np.uint8)

    def initUndistortRectifyMap(K, d, R, newK, size, m1type):
        return (size, None)

    def remap(img, mapx, mapy, interpolation=None):
        W, H = mapx
        return np.zeros((H, W, 3), np.uint8)

    cv2.circle = circle
    cv2.imread = imread
    cv2.initUndistortRectifyMap = initUndistortRectifyMap
    cv2.remap = remap
    sys.modules["cv2"] = cv2
    sys.modules["ffmpeg"] = types.ModuleType("ffmpeg")


def pack_instances(prefix, instances, out):
    """list[{"class","points"}] -> flat arrays under out[prefix_*]."""
    classes = [ins["class"] for ins in instances]
    counts = np.asarray([ins["points"].shape[0] for ins in instances], np.int64)
    if len(instances):
        pts = np.concatenate([np.ascontiguousarray(ins["points"]) for ins in instances], axis=0)
    else:
        pts = np.zeros((0, 3))
    out[prefix + "_classes"] = np.asarray(classes, dtype="U32")
    out[prefix + "_counts"] = counts
    out[prefix + "_points"] = pts
    out[prefix + "_dtype"] = np.asarray(str(pts.dtype))


def run_clip(tag, clip_kwargs, mutate=None):
    from cama_amd.synth import make_clip, DEFAULT_CAMA_CONFIGS
    rec = _Recorder()
    install_stubs(rec)
    bind_reference_package()
    sys.path.insert(0, REFERENCE)
    try:
        from cama.dataset import ClipManager
        from cama.dataset_reader import DatasetReader
        from cama.tools import VideoGenerator
        assert_from_reference(ClipManager, DatasetReader, VideoGenerator)
        tmp = tempfile.mkdtemp(prefix="golden_")
        clip = os.path.join(tmp, "clip")
        make_clip(clip, **clip_kwargs)
        if mutate is not None:
            mutate(clip)
        out = {"clip_kwargs": np.asarray(json.dumps(clip_kwargs)),
               "mutate": np.asarray(mutate.__name__ if mutate else "")}
        configs = dict(DEFAULT_CAMA_CONFIGS)
        cm = ClipManager(configs, clip)
        # calibration as the reference derived it
        for c in cm.cm_list:
            out[f"cal_{c.camera_name}_chassis2camera"] = np.asarray(c.chassis2camera)
            out[f"cal_{c.camera_name}_K"] = np.asarray(c.K)
            out[f"cal_{c.camera_name}_K_origin"] = np.asarray(c.K_origin)
            out[f"cal_{c.camera_name}_wh"] = np.asarray([c.width, c.height, c.width_origin, c.height_origin])
        datasets = [d for d in ("cama", "nuscenes") if d in cm.instance_maps]
        out["datasets"] = np.asarray(datasets, dtype="U16")
        for ds in datasets:
            pack_instances(f"{ds}_static", cm.instance_maps[ds], out)
            # pose track exactly as yield_frame builds it (dataset.py:78-87)
            dr = DatasetReader(clip)
            pt = cm.get_pt_cama(dr) if ds == "cama" else cm.get_pt_nuscenes(dr)
            out[f"{ds}_pose_abs"] = np.asarray(pt.absolute_transform)
            out[f"{ds}_pose_stamps"] = np.asarray(pt.timestamps)
            stamps = dr.get_sensor_timestamp(configs["camera_main"], sync=True)
            out[f"{ds}_frame_stamps"] = np.asarray(stamps)
            w2c = {}
            for idx in range(1, len(stamps)):
                try:
                    c2w = pt.seek_by_timestamp(stamps[idx], t_max_diff=0.5, interpolate=True).astype(np.float32)
                except RuntimeError:
                    continue
                w2c[idx] = np.linalg.inv(c2w)
            frame_ids = []
            for image_idx, instance_map in cm.yield_frame(ds):
                frame_ids.append(image_idx)
                key = f"{ds}_f{image_idx}"
                out[key + "_w2c"] = w2c[image_idx]
                pack_instances(key + "_crop", instance_map, out)
                maps_2d = cm.project_all_camera(instance_map)
                rec.calls.clear()
                for c in cm.cm_list:
                    pack_instances(f"{key}_{c.camera_name}_vu", maps_2d[c.camera_name], out)
                    n0 = len(rec.calls)
                    img = np.zeros((c.height, c.width, 3), np.uint8)
                    c.render_maps(img, maps_2d[c.camera_name])
                    calls = rec.calls[n0:]
                    assert all(k[2] == 2 and k[6] == -1 for k in calls)
                    assert all(k[7] == "int32" for k in calls), set(k[7] for k in calls)
                    arr = np.asarray([k[:2] + k[3:6] for k in calls], np.int32).reshape(-1, 5)
                    out[f"{key}_{c.camera_name}_circles"] = arr   # u, v, b, g, r  in draw order
            out[f"{ds}_frame_ids"] = np.asarray(frame_ids, np.int64)
        np.savez_compressed(os.path.join(HERE, f"clip_{tag}.npz"), **out)
        print(f"clip_{tag}.npz: {len(out)} arrays, datasets={datasets}, "
              f"frames={ {ds: out[ds + '_frame_ids'].tolist() for ds in datasets} }")
        shutil.rmtree(tmp)
    finally:
        sys.path.remove(REFERENCE)
        for m in [k for k in sys.modules if k == "cama" or k.startswith("cama.")]:
            del sys.modules[m]


def mutate_pose_gaps(clip):
    """Create (a) a gap > 0.5 s between two pose rows, (b) a track that ends early,
    so some frames raise RuntimeError and are skipped (dataset.py:93-96)."""
    for name in ("scmv_camera_front.txt", "wigo_offset_clip.txt"):
        p = os.path.join(clip, "odometry", name)
        rows = np.loadtxt(p)
        keep = np.ones(len(rows), bool)
        keep[3] = False            # gap of 1.0 s around frame index 2
        keep[-3:] = False          # last frames out of range
        np.savetxt(p, rows[keep])


def mutate_single_point_labels(clip):
    """Append 2-vertex labels SHORTER than 0.2 units (but >= 0.1): int(len / 0.1) == 1, so each densifies to exactly ONE
    point (cama/reproject.py:53-63,83-93).  A one-point instance makes every matmul of the per-frame path a
    (4,4)@(4,1) / (3,3)@(3,1) product, which numpy hands to BLAS gemv instead of gemm -- a different accumulation
    order than the k-ordered FMA chain all multi-point instances follow."""
    rng = np.random.default_rng(77)
    c, s = np.cos(0.3), np.sin(0.3)                 # make_clip: drive from world_anchor (-290, -280) heading 0.3 rad
    p = os.path.join(clip, "maps", "map_labels.json")
    if os.path.exists(p):
        labels = json.load(open(p))
        for k in range(40):
            a, l = rng.uniform(2.0, 45.0), rng.uniform(-7.0, 7.0)
            x, y = -290.0 + a * c - l * s, -280.0 + a * s + l * c
            px, py = (y + 300.0) / 0.1, (x + 300.0) / 0.1           # BEV pixels (reproject.py:36-40 inverted)
            ang = rng.uniform(0, 2 * np.pi)
            L = rng.uniform(0.105, 0.195)
            labels.append({"attrs": {"type": ["lane_marking", "Road_teeth", "Crosswalk_Line"][k % 3]},
                           "data": [[px, py], [px + L * np.cos(ang), py + L * np.sin(ang)]], "id": 9100 + k})
        json.dump(labels, open(p, "w"))
    p = os.path.join(clip, "maps", "map_nuscenes.json")
    if os.path.exists(p):
        labels = json.load(open(p))
        for k in range(40):
            x, y = rng.uniform(-25.0, 45.0), rng.uniform(-9.0, 9.0)   # metres, frame of the track's middle pose
            ang = rng.uniform(0, 2 * np.pi)
            L = rng.uniform(0.105, 0.195)
            labels.append({"attrs": {"type": ["Road_teeth", "lane_marking", "Stop_Line"][k % 3]},
                           "data": [[x, y], [x + L * np.cos(ang), y + L * np.sin(ang)]], "id": 9200 + k})
        json.dump(labels, open(p, "w"))


def run_pose_seek():
    rec = _Recorder()
    install_stubs(rec)
    bind_reference_package()
    sys.path.insert(0, REFERENCE)
    try:
        from cama.pose_transformer import PoseTransformer, SlerpTransform, invT
        assert_from_reference(PoseTransformer, invT)
        from scipy.spatial.transform import Rotation
        rng = np.random.default_rng(7)
        P = 9
        stamps = 100.0 + np.cumsum(np.r_[0.0, rng.uniform(0.2, 0.45, P - 1)])
        stamps[5:] += 0.4            # one gap > 0.5 s between rows 4 and 5
        quat = Rotation.from_rotvec(rng.normal(0, 0.4, (P, 3))).as_quat()
        xyz = np.cumsum(rng.normal(0, 1.0, (P, 3)), axis=0)
        tum = np.concatenate([stamps[:, None], xyz, quat], axis=1)
        ext = np.eye(4)
        ext[:3, :3] = Rotation.from_rotvec([0.1, -0.2, 0.3]).as_matrix()
        ext[:3, 3] = [0.5, -0.1, 1.2]
        out = {"tum": tum, "ext": ext}
        pt = PoseTransformer()
        pt.loadarray(tum)
        out["abs_loaded"] = np.asarray(pt.absolute_transform)
        out["rel_loaded"] = np.asarray(pt.relative_transform)
        pt.right_rotate(ext)
        out["abs_right_rotate"] = np.asarray(pt.absolute_transform)
        pt2 = PoseTransformer()
        pt2.loadarray(tum)
        pt2.normalize2center()
        out["abs_normalize2center"] = np.asarray(pt2.absolute_transform)
        out["invT_ext"] = invT(ext)
        out["slerp_03"] = SlerpTransform(out["abs_loaded"][1], out["abs_loaded"][2], 0.3)
        queries = [float(stamps[0]), float(stamps[3]), float(stamps[3] + 5e-10), float(stamps[0] - 5e-10),
                   float(stamps[0] - 1e-3), float(stamps[-1] + 1e-3), float(stamps[-1]),
                   float((stamps[1] + stamps[2]) / 2), float(stamps[2] + 0.01), float(stamps[4] + 0.1),
                   float(stamps[6] + 0.123), float(stamps[7] - 1e-7)]
        res, ok = [], []
        for q in queries:
            try:
                res.append(pt.seek_by_timestamp(q, t_max_diff=0.5, interpolate=True))
                ok.append(1)
            except RuntimeError:
                res.append(np.full((4, 4), np.nan))
                ok.append(0)
        out["queries"] = np.asarray(queries)
        out["seek_ok"] = np.asarray(ok, np.int8)
        out["seek_result"] = np.asarray(res)
        res, ok = [], []
        for q in queries:
            try:
                res.append(pt.seek_by_timestamp(q, t_max_diff=0.5, interpolate=False))
                ok.append(1)
            except RuntimeError:
                res.append(np.full((4, 4), np.nan))
                ok.append(0)
        out["seek_nearest_ok"] = np.asarray(ok, np.int8)
        out["seek_nearest_result"] = np.asarray(res)
        np.savez_compressed(os.path.join(HERE, "pose_seek.npz"), **out)
        print("pose_seek.npz: ok flags", out["seek_ok"].tolist(), out["seek_nearest_ok"].tolist())
    finally:
        sys.path.remove(REFERENCE)
        for m in [k for k in sys.modules if k == "cama" or k.startswith("cama.")]:
            del sys.modules[m]


def run_mosaic():
    rec = _Recorder()
    install_stubs(rec)
    bind_reference_package()
    sys.path.insert(0, REFERENCE)
    try:
        from cama.tools import VideoGenerator
        from cama_amd.synth import CAMERA_NAMES
        assert_from_reference(VideoGenerator)
        rng = np.random.default_rng(3)
        H, W = 6, 8
        imgs = {n: rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for n in CAMERA_NAMES}
        vg = object.__new__(VideoGenerator)
        mosaic = VideoGenerator.concate_image(vg, imgs)
        out = {"mosaic": mosaic}
        for n in CAMERA_NAMES:
            out["img_" + n] = imgs[n]
        np.savez_compressed(os.path.join(HERE, "mosaic.npz"), **out)
        print("mosaic.npz:", mosaic.shape)
    finally:
        sys.path.remove(REFERENCE)
        for m in [k for k in sys.modules if k == "cama" or k.startswith("cama.")]:
            del sys.modules[m]


def make_eval_trajectories(seed=11, P=1300):
    """Synthetic gt / pred TUM arrays for the PoseEvaluator fixtures (shared with tests: pure numpy + scipy)."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    t = 50.0 + 0.1 * np.arange(P) + rng.uniform(-0.004, 0.004, P)
    yaw = np.cumsum(rng.normal(0.0, 0.004, P)) + 0.3 * np.sin(np.arange(P) / 150.0)
    speed = 9.0 + np.sin(np.arange(P) / 40.0)
    xy = np.cumsum(np.stack([np.cos(yaw), np.sin(yaw)], 1) * (0.1 * speed)[:, None], axis=0)
    z = 0.5 * np.sin(np.arange(P) / 90.0)
    rot = Rotation.from_euler("zyx", np.stack([yaw, 0.02 * np.sin(np.arange(P) / 30.0), np.zeros(P)], 1))
    gt = np.concatenate([t[:, None], xy, z[:, None], rot.as_quat()], axis=1)
    # prediction: another sensor frame + scale drift + noise, stamps jittered, some frames missing
    S = Rotation.from_rotvec([0.05, -0.1, 0.7])
    keep = np.ones(P, bool)
    keep[rng.choice(P, 60, replace=False)] = False
    keep[:2] = True
    pos = 0.83 * S.apply(gt[:, 1:4]) + np.array([3.0, -2.0, 0.7]) + np.cumsum(rng.normal(0, 0.004, (P, 3)), axis=0)
    prot = S * rot * Rotation.from_rotvec(np.cumsum(rng.normal(0, 2e-4, (P, 3)), axis=0))
    pred = np.concatenate([(t + rng.uniform(-0.012, 0.012, P))[:, None], pos, prot.as_quat()], axis=1)[keep]
    return gt, pred


def run_pose_eval():
    rec = _Recorder()
    install_stubs(rec)
    sys.path.insert(0, REFERENCE)
    try:
        bind_reference_package()
        from cama.pose_evaluator import PoseEvaluator
        assert_from_reference(PoseEvaluator)
        gt, pred = make_eval_trajectories()
        out = {"gt": gt, "pred": pred}
        cases = [("7dof", 1.0, 0), ("6dof", 1.0, 0), ("scale", 1.0, 0), ("scale_7dof", 1.0, 0), ("None", 1.0, 0),
                 ("6dof", 1.2, 0), ("7dof", 1.0, 0.3)]
        names = []
        for k, (alignment, scale, offset) in enumerate(cases):
            pe = PoseEvaluator(alignment=alignment, scale=scale, offset=offset)
            p = pred.copy()
            if offset:
                p[:, 0] -= offset
            res = pe.eval(gt.copy(), p)
            tag = f"case{k}"
            names.append(f"{alignment}|{scale}|{offset}")
            out[tag + "_keys"] = np.array(list(res.keys()))
            for key, val in res.items():
                out[f"{tag}_{key}"] = np.asarray(val, dtype=np.float64)
            if scale != 1.0:
                out[tag + "_pred_after"] = p                 # load_poses scales x,y of the caller's array in place
            out[tag + "_n_poses"] = np.int64(len(pe.poses_pred))
            out[tag + "_pose_pred_last"] = pe.poses_pred[len(pe.poses_pred) - 1]
            out[tag + "_pose_gt_last"] = pe.poses_gt[len(pe.poses_gt) - 1]
            seq = pe.calc_sequence_errors(pe.poses_gt, pe.poses_pred)
            out[tag + "_seq_err_rows"] = np.int64(len(seq))
            if k in (0, 5):
                out[tag + "_seq_err"] = np.asarray(seq, dtype=np.float64)
            seg = pe.compute_segment_error(seq)
            out[tag + "_seg_lengths"] = np.array([l for l in pe.lengths if len(seg[l])], dtype=np.int64)
            out[tag + "_seg_err"] = np.asarray([seg[l] for l in pe.lengths if len(seg[l])], dtype=np.float64)
        out["case_names"] = np.array(names)
        # pieces
        pe = PoseEvaluator(alignment="7dof")
        m = pe.associate(pe.array2dict(gt), pe.array2dict(pred))
        out["assoc_matches"] = np.asarray(m, dtype=np.float64)
        pe_off = PoseEvaluator(alignment="7dof", max_t_diff=0.25, offset=0.07)
        out["assoc_matches_wide"] = np.asarray(pe_off.associate(pe_off.array2dict(gt[:200]), pe_off.array2dict(pred[:150])),
                                               dtype=np.float64)
        rng = np.random.default_rng(5)
        x = rng.normal(0, 10, (3, 57))
        from scipy.spatial.transform import Rotation
        y = 1.7 * Rotation.from_rotvec([0.3, 0.2, -0.9]).apply(x.T).T + np.array([[1.0], [2.0], [3.0]]) + rng.normal(0, 0.01, (3, 57))
        for ws in (False, True):
            r, t, c = pe.umeyama_alignment(x, y, ws)
            out[f"umeyama_{int(ws)}_r"], out[f"umeyama_{int(ws)}_t"], out[f"umeyama_{int(ws)}_c"] = r, t, np.float64(c)
        out["umeyama_x"], out["umeyama_y"] = x, y
        # reflection branch (det < 0): mirrored target
        ym = y.copy()
        ym[2] *= -1
        r, t, c = pe.umeyama_alignment(x, ym, True)
        out["umeyama_mirror_r"], out["umeyama_mirror_t"], out["umeyama_mirror_c"] = r, t, np.float64(c)
        poses = pe.quaternion2transform(gt[:40, 1:])
        out["q2t"] = np.stack([poses[i] for i in range(40)])
        out["traj_dist"] = np.asarray(pe.trajectory_distances(poses))
        E = np.linalg.inv(poses[3]) @ poses[17]
        out["err_terms"] = np.array([pe.rotation_error(E), pe.translation_error(E), *pe.rpy_error(E)])
        out["last_frame"] = np.array([pe.last_frame_from_segment_length(list(out["traj_dist"]), f, L)
                                      for f in (0, 5, 39) for L in (1.0, 10.0, 1000.0)], dtype=np.int64)
        # too few matches
        try:
            PoseEvaluator(alignment="7dof").eval(gt[:8].copy(), pred[:8].copy())
            out["few_matches_raises"] = np.int8(0)
        except RuntimeError:
            out["few_matches_raises"] = np.int8(1)
        try:
            PoseEvaluator(alignment="7dof", scale=1.1)
            out["bad_scale_raises"] = np.int8(0)
        except RuntimeError:
            out["bad_scale_raises"] = np.int8(1)
        np.savez_compressed(os.path.join(HERE, "pose_eval.npz"), **out)
        print("pose_eval.npz:", names, "seq_err rows", [int(out[f"case{k}_seq_err_rows"]) for k in range(len(cases))],
              "matches", len(m))
    finally:
        sys.path.remove(REFERENCE)
        for m_ in [k for k in sys.modules if k == "cama" or k.startswith("cama.")]:
            del sys.modules[m_]


def make_sensor_pack(root, legacy):
    """Tiny synthetic pack with lidar sweeps and IMU / GNSS / wheel logs (current dict-style frames, or the
    deprecated list-style ones when `legacy`), shared by the generator and the tests."""
    rng = np.random.default_rng(21 + int(legacy))
    os.makedirs(root, exist_ok=True)
    lidar = [1700000000000 + 100 * i for i in range(4)]
    fast = [1700000000000 + 33 * i for i in range(9)]
    attr = {"sync": {"lidar_top": lidar, "wheel": lidar}, "unsync": {"IMU": fast, "UB482": fast[::2], "wheel": fast},
            "calibration": {}}
    json.dump(attr, open(os.path.join(root, "attribute.json"), "w"))
    for d in ("lidar_top", "deskewed_lidar_top", "IMU", "UB482", "wheel"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for k, ts in enumerate(lidar):
        rng.normal(0, 20, (5 + k, 6)).tofile(os.path.join(root, "lidar_top", f"{ts}.bin"))
        rng.normal(0, 20, (5 + k, 6)).tofile(os.path.join(root, "deskewed_lidar_top", f"{ts}.bin"))
    imu = {str(ts): {"acc": rng.normal(0, 1, 3).tolist(), "gyro": rng.normal(0, 0.1, 3).tolist()} for ts in fast}
    json.dump(imu, open(os.path.join(root, "IMU", "data.json"), "w"))
    gnss, wheel = {}, {}
    for ts in fast:
        pos, quat = rng.normal(0, 100, 3).tolist(), rng.normal(0, 1, 4)
        quat = (quat / np.linalg.norm(quat)).tolist()
        if legacy:
            gnss[str(ts)] = {"position": pos, "orientation": quat}
            wheel[str(ts)] = {"x": rng.normal(), "y": rng.normal(), "z": rng.normal(), "roll": rng.normal(0, 0.05),
                              "pitch": rng.normal(0, 0.05), "yaw": rng.normal(0, 1.5)}
        else:
            gnss[str(ts)] = {"position": dict(zip("xyz", pos)), "orientation": dict(zip("xyzw", quat))}
            wheel[str(ts)] = {"x": rng.normal(), "y": rng.normal(), "yaw": rng.normal(0, 1.5)}
    for ts in lidar:
        wheel.setdefault(str(ts), dict(wheel[str(fast[0])]))
    json.dump(gnss, open(os.path.join(root, "UB482", "data.json"), "w"))
    json.dump(wheel, open(os.path.join(root, "wheel", "data.json"), "w"))
    return root


def run_dataset_reader():
    import warnings
    rec = _Recorder()
    install_stubs(rec)
    bind_reference_package()
    sys.path.insert(0, REFERENCE)
    tmp = tempfile.mkdtemp(prefix="golden_pack_")
    try:
        from cama.dataset_reader import DatasetReader
        assert_from_reference(DatasetReader)
        out = {}
        for legacy in (0, 1):
            dr = DatasetReader(make_sensor_pack(os.path.join(tmp, f"pack{legacy}"), bool(legacy)))
            tag = f"p{legacy}_"
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                out[tag + "gnss_tum"] = dr.get_GNSS_tum()
                out[tag + "wheel_tum_unsync"] = dr.get_wheel_tum()
                out[tag + "wheel_tum_sync"] = dr.get_wheel_tum(sync=True)
                out[tag + "n_warnings"] = np.int64(len(w))
            for deskewed in (0, 1):
                sweeps = list(dr.yield_lidar(start_idx=1, deskewed=bool(deskewed)))
                out[f"{tag}lidar{deskewed}_t"] = np.array([t for t, _ in sweeps])
                out[f"{tag}lidar{deskewed}_n"] = np.array([len(p) for _, p in sweeps])
                out[f"{tag}lidar{deskewed}_sum"] = np.array([p.sum() for _, p in sweeps])
            for name, it in (("imu", dr.yield_IMU()), ("gnss", dr.yield_GNSS()), ("wheel", dr.yield_wheel()),
                             ("wheel_unsync", dr.yield_wheel(sync=False))):
                frames = list(it)
                out[f"{tag}{name}_t"] = np.array([t for t, _ in frames])
                out[f"{tag}{name}_json"] = np.array(json.dumps([f for _, f in frames], sort_keys=True))
        np.savez_compressed(os.path.join(HERE, "dataset_reader.npz"), **out)
        print("dataset_reader.npz:", len(out), "arrays; warnings", int(out["p0_n_warnings"]), int(out["p1_n_warnings"]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        sys.path.remove(REFERENCE)
        _drop_cama_modules()


def main():
    if not os.path.isdir(REFERENCE):
        print("reference not present; golden vectors are committed, nothing to do")
        return 0
    sys.path.insert(0, REPO)
    only = {"pose_eval": run_pose_eval, "dataset_reader": run_dataset_reader,
            "f_single": lambda: run_clip("f_single", dict(n_frames=4, seed=5, n_lines=3, verts_per_line=4, line_len_m=2.0,
                                                           raster_size=400), mutate=mutate_single_point_labels)}
    if len(sys.argv) > 1 and sys.argv[1] in only:                 # regenerate one fixture only
        only[sys.argv[1]]()
        return 0
    run_clip("a", dict(n_frames=5, seed=0, n_lines=6, verts_per_line=5, line_len_m=2.0, raster_size=400))
    run_clip("b_exact", dict(n_frames=4, seed=1, n_lines=4, verts_per_line=4, line_len_m=1.5,
                             raster_size=300, pose_offset_s=0.0))
    run_clip("c_gaps", dict(n_frames=8, seed=2, n_lines=4, verts_per_line=3, line_len_m=1.0,
                            raster_size=300), mutate=mutate_pose_gaps)
    run_clip("d_nusonly", dict(n_frames=3, seed=3, n_lines=5, verts_per_line=6, line_len_m=1.0,
                               raster_size=300, with_cama=False))
    run_clip("e_crop", dict(n_frames=3, seed=4, n_lines=3, verts_per_line=4, line_len_m=14.0,
                            raster_size=300, with_cama=False))
    run_clip("f_single", dict(n_frames=4, seed=5, n_lines=3, verts_per_line=4, line_len_m=2.0, raster_size=400),
             mutate=mutate_single_point_labels)
    run_pose_seek()
    run_mosaic()
    run_pose_eval()
    run_dataset_reader()
    return 0


if __name__ == "__main__":
    sys.exit(main())
