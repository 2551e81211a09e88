"""Synthetic clip generator: writes a clip directory in the on-disk format the
reference demo consumes (SURVEY.md section 5 "clip format"; reference writer is
/root/reference/dataset/nuscenes2clip.py:467-522,646-704, readers are
cama/dataset_reader.py:29-43,150-168,278-294,409-411 and cama/dataset.py:26-76).

There is no nuScenes data on either box, so every test, golden vector and bench
input starts here.  Everything is seeded; nothing here is on the hot path.

Layout produced under <clip>/:
  attribute.json                      calibration + sync/unsync timestamp lists (ms ints)
  odometry/scmv_camera_front.txt      TUM rows, camera_front -> world   ("cama" pass)
  odometry/wigo_offset_clip.txt       TUM rows, chassis -> offset world ("nuscenes" pass)
  maps/map_labels.json                CAMA labels, BEV-pixel coordinates
  maps/vision_road_mlp_ft.npy         BEV height raster, [row=py, col=px]
  maps/map_nuscenes.json              nuScenes labels, metres, centre frame
  <camera>/<timestamp_ms>.jpg|.npy    frames (optional)
"""
import json
import os
from os.path import join

import numpy as np
from scipy.spatial.transform import Rotation

CAMERA_NAMES = ["camera_front_left", "camera_front", "camera_front_right",
                "camera_rear_left", "camera_rear", "camera_rear_right"]
CAMERA_YAW_DEG = {"camera_front_left": 55.0, "camera_front": 0.0, "camera_front_right": -55.0,
                  "camera_rear_left": 110.0, "camera_rear": 180.0, "camera_rear_right": -110.0}
MAP_CLASSES = ["lane_marking", "Road_teeth", "Crosswalk_Line"]
K_NUSCENES_LIKE = [[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]]

DEFAULT_CAMA_CONFIGS = {
    "result_dir": "maps",
    "camera_list": list(CAMERA_NAMES),
    "camera_main": "camera_front",
    "height_mlp": "vision_road_mlp_ft.npy",
    "pose_prefix": "scmv",
    "cama_map_file": "map_labels.json",
    "nuscenes_map_file": "map_nuscenes.json",
}


def camera_to_chassis(yaw_deg, mount=(1.5, 0.0, 1.5)):
    """4x4 camera->chassis.  Chassis: x fwd, y left, z up.  Camera: z fwd, x right, y down."""
    a = np.deg2rad(yaw_deg)
    fwd = np.array([np.cos(a), np.sin(a), 0.0])
    right = np.array([np.sin(a), -np.cos(a), 0.0])
    down = np.array([0.0, 0.0, -1.0])
    T = np.eye(4)
    T[:3, 0] = right
    T[:3, 1] = down
    T[:3, 2] = fwd
    T[:3, 3] = mount
    return T


def _tum_rows(stamps, T_list):
    rows = []
    for t, T in zip(stamps, T_list):
        q = Rotation.from_matrix(T[:3, :3]).as_quat()
        rows.append([t, T[0, 3], T[1, 3], T[2, 3], q[0], q[1], q[2], q[3]])
    return np.asarray(rows, dtype=np.float64)


def drive_track(n_poses, t0, dt, start_xy, speed_m, yaw0, yaw_rate, z=0.0):
    """chassis->world poses of a gentle arc: `speed_m` metres and `yaw_rate` rad per pose."""
    poses = []
    x, y, yaw = float(start_xy[0]), float(start_xy[1]), float(yaw0)
    for _ in range(n_poses):
        T = np.eye(4)
        T[:3, :3] = Rotation.from_euler("z", yaw).as_matrix()
        T[:3, 3] = (x, y, z)
        poses.append(T)
        x += speed_m * np.cos(yaw)
        y += speed_m * np.sin(yaw)
        yaw += yaw_rate
    stamps = t0 + dt * np.arange(n_poses)
    return stamps, poses


def lane_polylines(rng, n_lines, verts_per_line, line_len, origin_xy, heading,
                   along_span, lateral_span, wobble=0.15):
    """`n_lines` polylines of `verts_per_line` vertices, each `line_len` long, roughly parallel
    to `heading`, scattered over an along x lateral rectangle that starts at origin_xy."""
    c, s = np.cos(heading), np.sin(heading)
    lines = []
    lanes = max(1, int(np.sqrt(n_lines)))
    for i in range(n_lines):
        a0 = along_span[0] + (along_span[1] - along_span[0] - line_len) * rng.random()
        lat = lateral_span[0] + (lateral_span[1] - lateral_span[0]) * ((i % lanes) + 0.5) / lanes
        a = a0 + np.linspace(0.0, line_len, verts_per_line)
        l = lat + wobble * np.sin(a * 0.21 + rng.random() * 6.28)
        x = origin_xy[0] + a * c - l * s
        y = origin_xy[1] + a * s + l * c
        lines.append(np.stack([x, y], axis=-1))
    return lines


def make_clip(clip_path, n_frames=5, seed=0, n_lines=6, verts_per_line=5,
              line_len_m=2.0, raster_size=400, image_mode="none",
              image_size=(900, 1600), pose_offset_s=0.25, pose_dt_s=0.5,
              with_cama=True, with_nuscenes=True, nus_line_len_m=None,
              world_anchor=(-290.0, -280.0), extra_labels=True, d_nonzero=False, origin_size=(900, 1600)):
    """Write one synthetic clip; returns a dict describing it.

    n_frames counts *sync timestamps* (the demo renders indices 1..n_frames-1,
    cama/dataset.py:88).  Pose rows bracket every image stamp with `pose_dt_s`
    spacing shifted by `pose_offset_s` so each frame takes the interpolation
    branch (pose_transformer.py:630-642).  `raster_size` is the BEV raster side
    in pixels (0.1 m/px; the real one is 6000, reproject.py:23-27).
    """
    rng = np.random.default_rng(seed)
    os.makedirs(clip_path, exist_ok=True)
    for sub in ("odometry", "maps"):
        os.makedirs(join(clip_path, sub), exist_ok=True)

    # ---- timestamps (ms ints), 0.5 s frame spacing like nuScenes keyframes
    t0_ms = 1_600_000_000_000 + 1000 * seed
    frame_ms = [int(t0_ms + 500 * i) for i in range(n_frames)]
    frame_s = np.asarray(frame_ms, dtype=np.float64) / 1000.0

    # ---- chassis -> world track: poses at frame times shifted by pose_offset_s
    n_poses = int(np.ceil((frame_s[-1] - frame_s[0] + 2 * pose_dt_s) / pose_dt_s)) + 2
    speed = 2.0 * (pose_dt_s / 0.5)
    yaw0 = 0.3
    stamps, chassis2world = drive_track(
        n_poses, frame_s[0] - pose_dt_s + pose_offset_s, pose_dt_s,
        start_xy=world_anchor, speed_m=speed, yaw0=yaw0, yaw_rate=0.01 * (pose_dt_s / 0.5),
        z=0.02)

    # ---- calibration
    calibration = {}
    cam2chassis = {}
    for name in CAMERA_NAMES:
        T = camera_to_chassis(CAMERA_YAW_DEG[name])
        # small seeded perturbation so matrices carry full fp64 mantissas
        dR = Rotation.from_rotvec(rng.normal(0, 0.01, 3)).as_matrix()
        T[:3, :3] = T[:3, :3] @ dR
        T[:3, 3] += rng.normal(0, 0.05, 3)
        cam2chassis[name] = T
        calibration[f"{name}_2_chassis"] = T.tolist()
        K = np.array(K_NUSCENES_LIKE) + np.diag([rng.normal(0, 2.0), rng.normal(0, 2.0), 0.0])
        K[0, 2] += rng.normal(0, 3.0)
        K[1, 2] += rng.normal(0, 3.0)
        d = [0.0] * 8
        if d_nonzero:
            d = [-0.05, 0.01, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        if tuple(origin_size) != (900, 1600):       # small native resolution for cheap tests
            K[0] *= origin_size[1] / 1600.0
            K[1] *= origin_size[0] / 900.0
        calibration[name] = {
            "center_u": K[0, 2], "center_v": K[1, 2], "distort": d,
            "focal_u": K[0, 0], "focal_v": K[1, 1], "fov": 70.0,
            "image_height": int(origin_size[0]), "image_width": int(origin_size[1]), "K": K.tolist(), "d": d,
        }
    attribute = {
        "start_time": frame_ms[0], "end_time": frame_ms[-1], "status": "synthetic",
        "calibration": calibration,
        "unsync": {name: list(frame_ms) for name in CAMERA_NAMES},
        "sync": {name: list(frame_ms) for name in CAMERA_NAMES},
    }
    with open(join(clip_path, "attribute.json"), "w") as f:
        json.dump(attribute, f)

    # ---- odometry
    cam_main2chassis = cam2chassis["camera_front"]
    cam2world = [T @ cam_main2chassis for T in chassis2world]
    np.savetxt(join(clip_path, "odometry", "scmv_camera_front.txt"), _tum_rows(stamps, cam2world))
    mid = chassis2world[len(chassis2world) // 2][:3, 3].copy()
    offset = []
    for T in chassis2world:
        To = T.copy()
        To[:3, 3] -= mid
        offset.append(To)
    np.savetxt(join(clip_path, "odometry", "wigo_offset_clip.txt"), _tum_rows(stamps, offset))

    # ---- labels.  Drive starts at world_anchor heading yaw0 and covers ~2 m/frame.
    drive_len = speed * (n_frames + 2)
    info = {"clip_path": clip_path, "n_frames": n_frames, "frame_ms": frame_ms}
    if with_cama:
        lines = lane_polylines(rng, n_lines, verts_per_line, line_len_m, world_anchor, yaw0,
                               along_span=(-5.0, drive_len + 25.0), lateral_span=(-7.0, 7.0))
        labels = []
        for i, xy in enumerate(lines):
            # world x = py*0.1-300, world y = px*0.1-300  (reproject.py:36-40)  => invert
            px = (xy[:, 1] + 300.0) / 0.1
            py = (xy[:, 0] + 300.0) / 0.1
            data = np.stack([px, py], axis=-1)
            labels.append({"attrs": {"type": MAP_CLASSES[i % 3]}, "data": data.tolist(), "id": i})
        if extra_labels:
            labels.append({"attrs": {"type": "lane_marking"}, "data": [[10.0, 12.0]], "id": 9001})  # 1 vertex: dropped
            labels.append({"attrs": {"type": "Stop_Line"},                                         # short + negative px
                           "data": [[-3.2, 5.0], [-3.15, 5.02], [2.0, 7.5], [2.0, 7.5], [4.0, 9.0]], "id": 9002})
            labels.append({"attrs": {"type": "Road_teeth"},                                        # beyond raster edge
                           "data": [[raster_size - 2.0, raster_size - 3.0], [raster_size + 4.0, raster_size + 1.0]], "id": 9003})
        with open(join(clip_path, "maps", "map_labels.json"), "w") as f:
            json.dump(labels, f)
        raster = rng.normal(0.0, 0.05, (raster_size, raster_size)).astype(np.float32)
        np.save(join(clip_path, "maps", "vision_road_mlp_ft.npy"), raster)
        info["cama_labels"] = len(labels)
    if with_nuscenes:
        L = nus_line_len_m if nus_line_len_m is not None else line_len_m * 10.0
        centre = np.asarray(offset[len(offset) // 2])  # normalize2center frame == this pose's frame
        lines = lane_polylines(rng, n_lines, verts_per_line, L, (0.0, 0.0), 0.0,
                               along_span=(-drive_len / 2 - 10.0, drive_len / 2 + 40.0),
                               lateral_span=(-9.0, 9.0))
        labels = []
        for i, xy in enumerate(lines):
            labels.append({"attrs": {"type": MAP_CLASSES[(i + 1) % 3]}, "data": xy.tolist(), "id": i})
        if extra_labels:
            labels.append({"attrs": {"type": "lane_marking"}, "data": [], "id": 9004})
            labels.append({"attrs": {"type": "Crosswalk_Line"},                                    # one num==0 segment inside
                           "data": [[1.0, 1.0], [1.02, 1.03], [3.0, 2.0]], "id": 9005})
        with open(join(clip_path, "maps", "map_nuscenes.json"), "w") as f:
            json.dump(labels, f)
        info["nuscenes_labels"] = len(labels)
        del centre

    # ---- frames
    if image_mode != "none":
        H, W = image_size
        for ci, name in enumerate(CAMERA_NAMES):
            os.makedirs(join(clip_path, name), exist_ok=True)
            for fi, ts in enumerate(frame_ms):
                if image_mode == "jpg_photo":
                    # photo-like content (smooth structure + sensor noise): ~300 KB per 1600x900 JPEG, where pure
                    # noise ("jpg", the worst case for any JPEG decoder) gives ~1.3 MB.  Own generator: the draws of
                    # the other modes (golden fixtures) are untouched.
                    prng = np.random.default_rng([seed, ci, fi])
                    yy, xx = np.mgrid[0:H, 0:W]
                    base = np.stack([(xx * 0.16 + 20 * np.sin(yy / 30 + fi)) % 256, (yy * 0.28 + 7 * ci) % 256,
                                     ((xx + yy) * 0.1 + 3 * fi) % 256], -1)
                    img = np.clip(base + prng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)
                    from PIL import Image
                    Image.fromarray(img[:, :, ::-1]).save(join(clip_path, name, f"{ts}.jpg"), quality=90)
                    continue
                img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
                if image_mode == "npy":
                    np.save(join(clip_path, name, f"{ts}.npy"), img)
                elif image_mode == "jpg":
                    from PIL import Image
                    Image.fromarray(img[:, :, ::-1]).save(join(clip_path, name, f"{ts}.jpg"), quality=90)
                else:
                    raise ValueError(image_mode)
    return info


# ---------------------------------------------------------------------------------------------------------------
# camera-frame content that is identical on every device
# ---------------------------------------------------------------------------------------------------------------
def _pcg_words(counter, seed, xp):
    """PCG-RXS-M-XS style output of a 32-bit counter hash, in int64 arithmetic that never exceeds 2^62 (so numpy,
    torch-CPU and torch-GPU agree bit for bit): `counter` int64 array/tensor -> values in [0, 2^32)."""
    m32 = 0xFFFFFFFF
    h = (counter * 747796405 + (int(seed) * 2891336453 + 12345) % (1 << 31)) & m32
    h = (((h >> ((h >> 28) + 4)) ^ h) * 277803737) & m32
    return ((h >> 22) ^ h) & m32


def frame_pattern_np(seed, shape, first=0):
    """uint8 ndarray of `shape` (byte count a multiple of 4): byte stream = little-endian PCG words of counters
    first/4, first/4 + 1, ... of stream `seed`.  `first` = byte offset into the stream (a multiple of 4)."""
    n = int(np.prod(shape))
    assert n % 4 == 0 and first % 4 == 0
    c = np.arange(first // 4, first // 4 + n // 4, dtype=np.int64)
    return _pcg_words(c, seed, np).astype(np.uint32).view(np.uint8).reshape(shape)


def frame_pattern(seed, shape, device, chunk_bytes=1 << 28, first=0):
    """torch twin of frame_pattern_np on any device (filled in chunks: the int64 temporaries are 8x the output)."""
    import torch
    n = int(np.prod(shape))
    assert n % 4 == 0 and first % 4 == 0
    out = torch.empty(n, dtype=torch.uint8, device=device)
    words = out.view(torch.int32)
    for lo in range(0, n // 4, chunk_bytes // 4):
        hi = min(n // 4, lo + chunk_bytes // 4)
        c = torch.arange(first // 4 + lo, first // 4 + hi, dtype=torch.int64, device=device)
        w = _pcg_words(c, seed, torch)
        words[lo:hi] = (w - ((w >> 31) << 32)).to(torch.int32)       # [0, 2^32) -> the int32 with the same bits
    return out.view(shape)
