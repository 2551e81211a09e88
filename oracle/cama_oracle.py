"""oracle/cama_oracle.py -- TEST INFRASTRUCTURE ONLY.

numpy/scipy restatement ("port") of the reference's CPU path for the CAMA
reprojection hot path, written as plain functions.  It keeps the reference's
loop structure (per frame -> per camera -> per instance numpy calls -> per point
circle call) because that structure IS the CPU baseline bench.py times
(`cpu_baseline.kind == "port"`).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; cama_amd/ never does.

Pinned against tests/golden/*.npz, which were produced by importing the real
reference in the build container (tests/golden/gen_golden.py).
PARITY UNPINNED beyond the third-party boundaries, because neither OpenCV nor
ffmpeg is installed on either box: the filled-circle footprint
(oracle_circle_fill in cama_oracle.c), cv2.initUndistortRectifyMap
(oracle_undistort_map: OpenCV's published scalar loop, operation by operation),
cv2.remap (remap_bilinear) and libswscale's bgr24 -> yuv420p (bgr_to_i420) are
restated from the published algorithms.  tests/test_cv2_pins.py pins the three
OpenCV ones against the real library wherever `import cv2` works; cv2.imread's
decode IS pinned (oracle/jpeg_oracle.py == libjpeg-turbo through Pillow).

Each function cites the reference lines it follows (paths under /root/reference).
"""
import ctypes
import json
import os
from os.path import join

import numpy as np
from scipy.spatial.transform import Rotation, Slerp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

GREY_RGB = (211, 211, 211)      # cama/reproject.py:14  "lane_marking"
GOLD_RGB = (255, 215, 0)        # cama/reproject.py:16  "Crosswalk_Line" (used for every other class, :251-252)
CROP_BOX = (-50.0, 50.0, -100.0, 100.0, -200.0, 200.0)   # cama/reproject.py:28-34
STEP = 0.1                      # cama/reproject.py:23
MAP_EXTENT = 600.0              # cama/reproject.py:26-27
MOSAIC_ORDER = ["camera_front_left", "camera_front", "camera_front_right",
                "camera_rear_left", "camera_rear", "camera_rear_right"]   # cama/tools.py:23-24


def lib():
    """ctypes handle of oracle/_build/liboracle.so (make -C oracle)."""
    global _LIB
    if _LIB is None:
        path = join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        L.oracle_frame_project.argtypes = [vp, i32, i64, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
        L.oracle_frame_project.restype = None
        L.oracle_transform.argtypes = [vp, i64, vp, vp]
        L.oracle_transform.restype = None
        L.oracle_circle_fill.argtypes = [vp, i32, i32, i64, i32, i32, i32, i32, i32, i32]
        L.oracle_circle_fill.restype = None
        L.oracle_line_bresenham.argtypes = [vp, i32, i32, i64, i32, i32, i32, i32, i32, i32, i32]
        L.oracle_line_bresenham.restype = None
        L.oracle_circle_halfwidths.argtypes = [i32, vp]
        L.oracle_circle_halfwidths.restype = i32
        L.oracle_render_frame.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i64, i32, vp]
        L.oracle_render_frame.restype = None
        L.oracle_render_frame_alpha.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, i64, i32, vp, i32, vp]
        L.oracle_render_frame_alpha.restype = None
        L.oracle_render_frame_wu.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, i64, i32, vp, vp]
        L.oracle_render_frame_wu.restype = None
        L.oracle_undistort_map.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
        L.oracle_undistort_map.restype = i32
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- static map
def densify(line_points):
    """cama/reproject.py:51-63 == :81-93.  line_points: (k,2) float32, k >= 2."""
    out = []
    seg_len = np.linalg.norm(line_points[1:] - line_points[:-1], axis=-1)
    for s in range(len(seg_len)):
        a = line_points[s]
        b = line_points[s + 1]
        num = int(seg_len[s] / STEP)
        if num == 0:
            continue
        for j in range(num):
            out.append(a + (b - a) / num * j)
    return np.array(out)


def pixel_to_world_xy(pixel_xy):
    """cama/reproject.py:36-40 (note the axis swap; centre is (0,0))."""
    w = np.zeros_like(pixel_xy)
    w[:, 0] = pixel_xy[:, 1] * STEP - MAP_EXTENT / 2 + 0
    w[:, 1] = pixel_xy[:, 0] * STEP - MAP_EXTENT / 2 + 0
    return w


def static_map_cama(bev_height, labels):
    """cama/reproject.py:72-106."""
    result = []
    for label in labels:
        pts = label["data"]
        if len(pts) <= 1:
            continue
        dense = densify(np.array(pts).astype(np.float32))
        pix = dense.round().astype(np.uint16)
        pix = pix[:, ::-1]
        pix = pix.clip(0, bev_height.shape[0] - 1)
        h = bev_height[pix[:, 0], pix[:, 1]]
        xy = pixel_to_world_xy(dense)
        xyz = np.concatenate((xy, h[:, None]), axis=-1).reshape(-1, 3)
        result.append({"class": label["attrs"]["type"], "points": xyz})
    return result


def static_map_nuscenes(labels):
    """cama/reproject.py:42-70."""
    result = []
    for label in labels:
        pts = label["data"]
        if len(pts) <= 1:
            continue
        dense = densify(np.array(pts).astype(np.float32))
        h = np.zeros_like(dense[:, 0])
        xyz = np.concatenate((dense, h[:, None]), axis=-1).reshape(-1, 3)
        result.append({"class": label["attrs"]["type"], "points": xyz})
    return result


# --------------------------------------------------------------------------- pose track
def inv_rigid(T):
    """cama/pose_transformer.py:8-21."""
    Rt = T[:3, :3].T
    out = np.eye(4)
    out[:3, :3] = Rt
    out[:3, 3] = -Rt @ T[:3, 3]
    return out


def tum_to_poses(tum):
    """cama/pose_transformer.py:429-438: rows t x y z qx qy qz qw -> (stamps (P,1), list of 4x4)."""
    assert tum.shape[1] == 8
    P = tum.shape[0]
    M = np.zeros((P, 4, 4))
    M[:, 3, 3] = 1
    M[:, :3, :3] = Rotation.from_quat(tum[:, 4:8]).as_matrix()
    M[:, :3, 3] = tum[:, 1:4]
    return tum[:, 0:1], list(M)


def right_compose(poses, ext):
    """cama/pose_transformer.py:520-537."""
    return [T @ ext for T in poses]


def normalize_to_center(poses):
    """cama/pose_transformer.py:324-336."""
    ref_inv = inv_rigid(poses[len(poses) // 2])
    return [ref_inv @ T for T in poses]


def slerp_transform(left, right, ratio):
    """cama/pose_transformer.py:24-44."""
    assert 0 <= ratio <= 1
    keys = Rotation.from_matrix(np.concatenate([left[np.newaxis, :3, :3], right[np.newaxis, :3, :3]], axis=0))
    rot = Slerp([0, 1], keys)(ratio).as_matrix()
    out = left * (1 - ratio) + right * ratio
    out[:3, :3] = rot
    return out


def seek_pose(stamps, poses, query, max_diff, interpolate):
    """cama/pose_transformer.py:589-652.  stamps: (P,1).  Raises RuntimeError like the reference."""
    assert isinstance(query, float) and isinstance(max_diff, float)
    if len(poses) == 0:
        raise RuntimeError("No poses found, pleas load poses first")
    if stamps.shape[0] == 0:
        raise RuntimeError("No timestamps found, pleas load timestamps first")
    assert np.all(stamps[1:, 0] >= stamps[:-1, 0]), "timestamps must be sorted"
    hit = np.where(np.isclose(stamps[:, 0], query, rtol=1e-20, atol=1e-9))[0]
    if hit.size > 0:
        return poses[hit[0]]
    right = np.searchsorted(stamps[:, 0], query, side="left")
    left = right - 1
    if interpolate:
        if right >= stamps.shape[0]:
            raise RuntimeError("query_time is out of range.")
        if right == 0 and -1e-9 < (query - stamps[0]) < 0:
            right, left = 1, 0
        elif query - stamps[0] < -1e-9:
            raise RuntimeError("query_time is out of range.")
        gap = stamps[right] - stamps[left]
        if gap > max_diff:
            raise RuntimeError(f"time_diff = {gap} is greater than t_max_diff {max_diff}")
        ratio = (query - stamps[left]) / gap
        return slerp_transform(poses[left], poses[right], ratio)
    dl = query - stamps[left] if left >= 0 else float("inf")
    dr = stamps[right] - query if right < stamps.shape[0] else float("inf")
    d = min(dl, dr)[0]
    if d > max_diff:
        raise RuntimeError(f"time_diff = {d} is greater than t_max_diff {max_diff}")
    return poses[left if dl < dr else right]


# --------------------------------------------------------------------------- clip reader subset
def read_attribute(clip_path):
    """cama/dataset_reader.py:29-37."""
    p = join(clip_path, "attribute.json")
    if not os.path.exists(p):
        raise FileNotFoundError("can not find {}".format(p))
    with open(p, "r") as f:
        return json.load(f)


def sensor_seconds(attribute, sensor, sync=True):
    """cama/dataset_reader.py:39-43."""
    ts = np.asarray(attribute["sync" if sync else "unsync"][sensor]).astype(np.double)
    ts /= 1000.0
    return ts.tolist()


def _direct_extrinsic(cal, a, b):
    """cama/dataset_reader.py:150-168."""
    if a == b:
        return np.eye(4, dtype=np.float32)
    if f"{a}_2_{b}" in cal:
        return np.asarray(cal[f"{a}_2_{b}"])
    if f"{b}_2_{a}" in cal:
        return inv_rigid(np.asarray(cal[f"{b}_2_{a}"]))
    return None


def extrinsic(attribute, a, b):
    """cama/dataset_reader.py:170-248 (direct, inverse, else BFS shortest path over the calibration graph)."""
    cal = attribute["calibration"]
    T = _direct_extrinsic(cal, a, b)
    if T is not None:
        return T
    graph = {}
    for key in cal:
        if "_2_" in key:
            s, d = key.split("_2_")
            graph.setdefault(s, []).append(d)
            graph.setdefault(d, []).append(s)
    seen, queue, path = [], [[a]], None
    while queue and path is None:
        cur = queue.pop(0)
        node = cur[-1]
        if node in seen:
            continue
        for nb in graph.get(node, []):
            nxt = cur + [nb]
            queue.append(nxt)
            if nb == b:
                path = nxt
                break
        seen.append(node)
    if path is None:
        print("extrinsic path not found!")
        return None
    T = np.eye(4, dtype=np.float32)
    for i in range(len(path) - 1):
        T = _direct_extrinsic(cal, path[i], path[i + 1]) @ T
    return T


def camera_model(attribute, name, output_size=(540, 960)):
    """cama/reproject.py:164-182: chassis->camera, K scaled to the output size."""
    intr = attribute["calibration"][name]
    K0 = np.asarray(intr.get("K"))
    W0, H0 = intr.get("image_width"), intr.get("image_height")
    H, W = output_size
    K = K0.copy()
    K[0, :] = K[0, :] * W / W0
    K[1, :] = K[1, :] * H / H0
    return {"name": name, "chassis2camera": extrinsic(attribute, "chassis", name),
            "K": K, "K_origin": K0, "W": W, "H": H, "W0": W0, "H0": H0,
            "d_origin": np.asarray(intr.get("d"))}


# --------------------------------------------------------------------------- per-frame geometry
def transform_instances(instances, T):
    """cama/reproject.py:108-116."""
    out = []
    for ins in instances:
        p = ins["points"]
        p = np.concatenate((p, np.ones((p.shape[0], 1))), axis=-1)
        p = (T @ p.T).T
        out.append({"class": ins["class"], "points": p[:, :3]})
    return out


def crop_instances(instances, box=CROP_BOX):
    """cama/reproject.py:118-131."""
    out = []
    for ins in instances:
        p = ins["points"]
        m = (p[:, 0] >= box[0]) & (p[:, 0] <= box[1]) & (p[:, 1] >= box[2]) & (p[:, 1] <= box[3]) & \
            (p[:, 2] >= box[4]) & (p[:, 2] <= box[5])
        p = p[m]
        if p.shape[0] > 0:
            out.append({"class": ins["class"], "points": p})
    return out


def project_instances(instances, K, W, H):
    """cama/reproject.py:187-205."""
    out = []
    for ins in instances:
        p = (K @ ins["points"].T).T
        front = p[:, 2] > 0
        p = p[:, :] / p[:, 2:]
        m = (p[:, 2] > 0) & (p[:, 0] >= 0) & (p[:, 0] < W) & (p[:, 1] >= 0) & (p[:, 1] < H)
        m = m & front
        p = p[m]
        if p.shape[0] > 0:
            out.append({"class": ins["class"], "points": p[:, :2][:, ::-1]})
    return out


def pose_track(clip_path, attribute, configs, dataset):
    """cama/dataset.py:60-76."""
    if dataset == "cama":
        main = configs["camera_main"]
        tum = np.loadtxt(join(clip_path, "odometry", f"{configs['pose_prefix']}_{main}.txt"))
        stamps, poses = tum_to_poses(tum)
        return stamps, right_compose(poses, extrinsic(attribute, "chassis", main))
    tum = np.loadtxt(join(clip_path, "odometry", "wigo_offset_clip.txt"))
    stamps, poses = tum_to_poses(tum)
    return stamps, normalize_to_center(poses)


def frame_world2chassis(stamps, poses, t):
    """cama/dataset.py:91-99: fp64 seek -> float32 -> float32 general inverse.  May raise RuntimeError."""
    c2w = seek_pose(stamps, poses, t, 0.5, True).astype(np.float32)
    return np.linalg.inv(c2w)


def iter_frames(clip_path, attribute, configs, static_map, dataset):
    """cama/dataset.py:78-106: yields (image_idx, world2chassis, cropped chassis-frame instances)."""
    stamps, poses = pose_track(clip_path, attribute, configs, dataset)
    secs = sensor_seconds(attribute, configs["camera_main"], sync=True)
    for idx in range(1, len(secs)):
        try:
            w2c = frame_world2chassis(stamps, poses, secs[idx])
        except RuntimeError:
            continue
        yield idx, w2c, crop_instances(transform_instances(static_map, w2c))


def project_all(cropped, cams):
    """cama/dataset.py:108-117."""
    return {c["name"]: project_instances(transform_instances(cropped, c["chassis2camera"]), c["K"], c["W"], c["H"])
            for c in cams}


# --------------------------------------------------------------------------- raster
def colour_bgr(class_name):
    """cama/reproject.py:250-254: everything but lane_marking is drawn in Crosswalk_Line gold; RGB -> BGR."""
    rgb = GREY_RGB if class_name == "lane_marking" else GOLD_RGB
    return tuple(int(v) for v in rgb[::-1])


def render_instances(image, maps_2d, radius=2, segments=False):
    """cama/reproject.py:246-257 with cv2.circle replaced by the C restatement; per-point Python loop kept.
    segments=True is the restatement of the product's opt-in EXTENSION (no reference counterpart): before point k's disc,
    a one-pixel Bresenham segment from point k - 1 to point k of the same instance, in point k's colour, when the two are
    neighbours (ins["joined"][k], default: every point but the instance's first)."""
    L = lib()
    H, W = image.shape[:2]
    assert image.dtype == np.uint8 and image.flags.c_contiguous
    if isinstance(segments, str) and segments.lower() == "wu":
        # the anti-aliased variant: the flat list of the image's points through oracle_render_frame_wu (one camera, all visible)
        pts = [ins["points"].astype(np.int32) for ins in maps_2d if len(ins["points"])]
        if not pts:
            return image
        link = np.concatenate([np.asarray(ins.get("joined", np.arange(len(ins["points"])) > 0), bool)
                               for ins in maps_2d if len(ins["points"])])
        col = np.concatenate([np.full(len(ins["points"]), 0 if ins["class"] == "lane_marking" else 1, np.uint8)
                              for ins in maps_2d if len(ins["points"])])
        vu = np.concatenate(pts).astype(np.float64)[None]
        image[...] = frame_render_flat_wu(image[None], vu, np.ones((1, vu.shape[1]), np.uint8), col, link, radius=radius, cols=1)
        return image
    base = image.ctypes.data
    step = image.strides[0]
    for ins in maps_2d:
        pts = ins["points"].astype(np.int32)
        b, g, r = colour_bgr(ins["class"])
        joined = np.asarray(ins.get("joined", np.arange(len(pts)) > 0), bool) if segments else None
        for k, p in enumerate(pts):
            if segments and k > 0 and joined[k]:
                L.oracle_line_bresenham(base, H, W, step, int(pts[k - 1][1]), int(pts[k - 1][0]), int(p[1]), int(p[0]), b, g, r)
            L.oracle_circle_fill(base, H, W, step, int(p[1]), int(p[0]), radius, b, g, r)
    return image


def mosaic(image_dict):
    """cama/tools.py:22-25."""
    top = np.concatenate([image_dict[n] for n in MOSAIC_ORDER[:3]], axis=1)
    bottom = np.concatenate([image_dict[n] for n in MOSAIC_ORDER[3:]], axis=1)
    return np.concatenate([top, bottom], axis=0)


def circle_halfwidths(radius):
    hw = np.zeros(radius + 1, np.int32)
    lib().oracle_circle_halfwidths(radius, _ptr(hw))
    return hw


# --------------------------------------------------------------------------- frame resample (OpenCV restated)
def undistort_map(K_origin, dist, K_new, W, H):
    """cv2.initUndistortRectifyMap(K_origin, dist, None, K_new, (W,H), CV_32FC1) as called at
    cama/reproject.py:238: oracle_undistort_map (cama_oracle.c) restates OpenCV's published scalar loop operation by
    operation -- closed-form 3x3 inverse of K_new, row-incremental _x/_y/_w, w = 1./_w, u = fx*invProj*xd + u0 (no
    skew), float64 arithmetic, float32 result.  PARITY UNPINNED (OpenCV absent; tests/test_cv2_pins.py pins it
    wherever cv2 is importable).  Non-zero tilt coefficients (dist[12:14]) raise."""
    K0 = np.ascontiguousarray(np.asarray(K_origin, np.float64).reshape(3, 3))
    Kn = np.ascontiguousarray(np.asarray(K_new, np.float64).reshape(3, 3))
    d = np.ascontiguousarray(np.asarray(dist if dist is not None else [], np.float64).reshape(-1))
    mapx = np.zeros((H, W), np.float32)
    mapy = np.zeros((H, W), np.float32)
    rc = lib().oracle_undistort_map(_ptr(K0), _ptr(d) if d.size else None, int(d.size), _ptr(Kn), int(W), int(H),
                                    _ptr(mapx), _ptr(mapy))
    if rc == -1:
        raise NotImplementedError("tilted-sensor coefficients (tauX, tauY) are not restated")
    if rc:
        raise ValueError("singular new camera matrix")
    return mapx, mapy


def remap_bilinear(image, mapx, mapy):
    """cv2.remap(image, mapx, mapy, INTER_LINEAR) for 8-bit images (cama/reproject.py:239), restated from OpenCV's
    fixed-point path: coordinates cvRound(map * 32) (INTER_BITS = 5), tap (x >> 5, y >> 5), weights
    (32-a)(32-b)*32 etc. out of 1 << 15 (INTER_REMAP_COEF_BITS = 15), rounding + (1 << 14) >> 15,
    BORDER_CONSTANT 0.  PARITY UNPINNED (OpenCV absent).

    Why exact integers are the right restatement: OpenCV's weight table (initInterTab2D, fixpt) is
    saturate_cast<short>(w * 32768) of float products of multiples of 1/32 -- exact -- and sums to 32768 for every
    fraction except (0,0), where the single weight 1.0 saturates to 32767 and the table's sum fix-up gives the missing
    1 to another tap; (32767*p00 + p11 + 16384) >> 15 still equals p00 for 8-bit pixels, the same value as the exact
    weights produce."""
    H0, W0 = image.shape[:2]
    sx = np.rint(mapx.astype(np.float32) * np.float32(32)).astype(np.int64)
    sy = np.rint(mapy.astype(np.float32) * np.float32(32)).astype(np.int64)
    x0, y0, a, b = sx >> 5, sy >> 5, sx & 31, sy & 31
    img = image.astype(np.int64)

    def tap(y, x):
        ok = (x >= 0) & (x < W0) & (y >= 0) & (y < H0)
        v = img[np.clip(y, 0, H0 - 1), np.clip(x, 0, W0 - 1)]
        return v * ok[..., None]
    w00, w01, w10, w11 = (32 - a) * (32 - b) * 32, a * (32 - b) * 32, (32 - a) * b * 32, a * b * 32
    acc = (w00[..., None] * tap(y0, x0) + w01[..., None] * tap(y0, x0 + 1) +
           w10[..., None] * tap(y0 + 1, x0) + w11[..., None] * tap(y0 + 1, x0 + 1) + (1 << 14)) >> 15
    return np.clip(acc, 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------- mosaic egress (libswscale restated)
I420_COEFFS = dict(ry=8414, gy=16519, by=3208, ru=-4865, gu=-9528, bu=14392, rv=14392, gv=-12061, bv=-2332)


def bgr_to_i420(image):
    """What the encoder ends up with for a bgr24 frame: VideoGenerator pipes raw bgr24 into ffmpeg with `-pix_fmt yuv420p`
    (cama/tools.py:13-20), i.e. libswscale converts BGR24 -> YUV420P.  Restated from libswscale's unscaled C converter
    (rgb2rgb_template.c rgb24toyv12_c, reached through bgr24ToYv12Wrapper for even widths without SWS_ACCURATE_RND):
    BT.601 limited range, coefficients int(c * 219|224 / 255 * 2^15 + 0.5), Y = ((ry*r + gy*g + by*b) >> 15) + 16 for
    every pixel, U / V = ((..) >> 15) + 128 (arithmetic shift) from the FIRST pixel of the FIRST line of each 2x2 block.
    Returns the planar I420 bytes (Y, U, V) as one uint8 vector.  PARITY UNPINNED: ffmpeg is absent on both boxes, and
    x86 builds may run a SIMD body that averages chroma instead."""
    H, W = image.shape[:2]
    assert H % 2 == 0 and W % 2 == 0 and image.dtype == np.uint8 and image.shape[2] == 3
    c = I420_COEFFS
    b, g, r = (image[..., k].astype(np.int64) for k in range(3))
    y = ((c["ry"] * r + c["gy"] * g + c["by"] * b) >> 15) + 16
    b0, g0, r0 = b[0::2, 0::2], g[0::2, 0::2], r[0::2, 0::2]
    u = ((c["ru"] * r0 + c["gu"] * g0 + c["bu"] * b0) >> 15) + 128
    v = ((c["rv"] * r0 + c["gv"] * g0 + c["bv"] * b0) >> 15) + 128
    return np.concatenate([y.astype(np.uint8).reshape(-1), u.astype(np.uint8).reshape(-1), v.astype(np.uint8).reshape(-1)])


# --------------------------------------------------------------------------- flat (C) frame path
def flatten_instances(instances):
    """-> (xyz (N,3) contiguous, colour_id (N,) uint8, counts, classes)."""
    classes = [ins["class"] for ins in instances]
    counts = np.asarray([ins["points"].shape[0] for ins in instances], np.int64)
    xyz = np.ascontiguousarray(np.concatenate([ins["points"] for ins in instances], axis=0)) if instances \
        else np.zeros((0, 3), np.float32)
    col = np.repeat(np.asarray([0 if c == "lane_marking" else 1 for c in classes], np.uint8), counts)
    return xyz, col, counts, classes


def frame_project_flat(xyz, w2c, cams, W, H, crop=CROP_BOX, want_chassis=False):
    """FMA-chain C path over a flat point buffer: returns dict(vu (C,N,2), vis (C,N), crop_mask (N,), chassis)."""
    L = lib()
    N, C = xyz.shape[0], len(cams)
    assert xyz.flags.c_contiguous and xyz.dtype in (np.float32, np.float64)
    w2c64 = np.ascontiguousarray(w2c, dtype=np.float64)
    c2cam = np.ascontiguousarray(np.stack([np.asarray(c["chassis2camera"], np.float64) for c in cams]))
    K = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float64) for c in cams]))
    cropa = np.asarray(crop, np.float64)
    vu = np.full((C, N, 2), np.nan)
    vis = np.zeros((C, N), np.uint8)
    cmask = np.zeros(N, np.uint8)
    chassis = np.zeros((N, 3)) if want_chassis else None
    L.oracle_frame_project(_ptr(xyz), int(xyz.dtype == np.float64), N, _ptr(w2c64), _ptr(cropa),
                           _ptr(c2cam), _ptr(K), C, W, H, _ptr(chassis), _ptr(cmask), _ptr(vu), _ptr(vis))
    return {"vu": vu, "vis": vis, "crop_mask": cmask, "chassis": chassis}


def frame_render_flat_segments(src, vu, vis, colour_id, link, radius=2, cols=3):
    """Restatement of the product's opt-in segment EXTENSION on a flat point buffer (no reference counterpart: the reference
    draws one disc per point, cama/reproject.py:255-256): per camera, in draw order, point k first gets the one-pixel
    Bresenham segment from point k - 1 (oracle_line_bresenham) when link[k] is set and BOTH points are visible in that
    camera, then its disc, all in point k's colour; later draws overwrite.  src (C,H,W,3) -> mosaic."""
    L = lib()
    C, H, W = src.shape[:3]
    rows = (C + cols - 1) // cols
    out = np.zeros((rows * H, cols * W, 3), np.uint8)
    pal = (tuple(int(v) for v in GREY_RGB[::-1]), tuple(int(v) for v in GOLD_RGB[::-1]))
    link = np.asarray(link, bool)
    for c in range(C):
        img = np.ascontiguousarray(src[c]).copy()
        base, step = img.ctypes.data, img.strides[0]
        ok = np.asarray(vis[c], bool)
        px = np.zeros((vu.shape[1], 2), np.int64)
        px[ok] = vu[c][ok].astype(np.int32)                         # reproject.py:249: truncation (values >= 0)
        for k in np.flatnonzero(ok):
            b, g, r = pal[int(colour_id[k]) & 1]
            if k > 0 and link[k] and ok[k - 1]:
                L.oracle_line_bresenham(base, H, W, step, int(px[k - 1][1]), int(px[k - 1][0]), int(px[k][1]), int(px[k][0]),
                                        b, g, r)
            L.oracle_circle_fill(base, H, W, step, int(px[k][1]), int(px[k][0]), radius, b, g, r)
        r0, q0 = divmod(c, cols)
        out[r0 * H:(r0 + 1) * H, q0 * W:(q0 + 1) * W] = img
    return out


def frame_render_flat_wu(src, vu, vis, colour_id, link, radius=2, cols=3):
    """Restatement of the product's opt-in ANTI-ALIASED segment extension (oracle_render_frame_wu in cama_oracle.c holds the
    definition: discs with coverage 255, Wu lines between linked visible neighbours with 8-bit coverages, per pixel the claim
    with the greatest (draw index, coverage), blended once over the source).  No reference counterpart.
    src (C,H,W,3) -> mosaic."""
    L = lib()
    C, H, W = src.shape[:3]
    N = vis.shape[1]
    rows = (C + cols - 1) // cols
    out = np.zeros((rows * H, cols * W, 3), np.uint8)
    pal = np.asarray([GREY_RGB[::-1], GOLD_RGB[::-1]], np.uint8)
    claim = np.zeros((H, W), np.uint32)
    L.oracle_render_frame_wu(_ptr(np.ascontiguousarray(src)), _ptr(out), C, H, W, cols, _ptr(np.ascontiguousarray(vu)),
                             _ptr(np.ascontiguousarray(vis).astype(np.uint8)), _ptr(np.ascontiguousarray(colour_id).astype(np.uint8) & 1),
                             _ptr(np.ascontiguousarray(link).astype(np.uint8)), N, radius, _ptr(pal), _ptr(claim))
    return out


def frame_render_flat(src, vu, vis, colour_id, radius=2, cols=3, alpha256=256):
    """src (C,H,W,3) uint8 -> mosaic ((C/cols)*H, cols*W, 3) through the C renderer.  alpha256 < 256 selects the
    translucent EXTENSION (own restatement, no reference semantics)."""
    L = lib()
    C, H, W = src.shape[:3]
    N = vis.shape[1]
    rows = (C + cols - 1) // cols
    out = np.zeros((rows * H, cols * W, 3), np.uint8)
    pal = np.asarray([GREY_RGB[::-1], GOLD_RGB[::-1]], np.uint8)
    if alpha256 != 256:
        layer = np.zeros((H, W, 3), np.uint8)
        L.oracle_render_frame_alpha(_ptr(np.ascontiguousarray(src)), _ptr(out), C, H, W, cols,
                                    _ptr(np.ascontiguousarray(vu)), _ptr(np.ascontiguousarray(vis)),
                                    _ptr(np.ascontiguousarray(colour_id)), N, radius, _ptr(pal), int(alpha256), _ptr(layer))
        return out
    L.oracle_render_frame(_ptr(np.ascontiguousarray(src)), _ptr(out), C, H, W, cols,
                          _ptr(np.ascontiguousarray(vu)), _ptr(np.ascontiguousarray(vis)),
                          _ptr(np.ascontiguousarray(colour_id)), N, radius, _ptr(pal))
    return out
