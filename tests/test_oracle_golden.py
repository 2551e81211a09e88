"""The oracle (oracle/cama_oracle.{py,c}) against the golden vectors captured from the real
reference (tests/golden/gen_golden.py).  CPU only.  This is what pins the oracle."""
import json
import os
from os.path import join

import numpy as np
import pytest

from oracle import cama_oracle as O
from tests.helpers import (CLIP_TAGS, CAMERA_NAMES, DEFAULT_CAMA_CONFIGS, GOLDEN, assert_instances_equal,
                           golden_instances, load_golden, rebuild_clip)


def _static_maps(clip):
    maps = {}
    try:
        labels = json.load(open(join(clip, "maps", "map_labels.json")))
        bev = np.load(join(clip, "maps", "vision_road_mlp_ft.npy"))
        maps["cama"] = O.static_map_cama(bev, labels)
    except FileNotFoundError:
        pass
    try:
        maps["nuscenes"] = O.static_map_nuscenes(json.load(open(join(clip, "maps", "map_nuscenes.json"))))
    except FileNotFoundError:
        pass
    return maps


@pytest.mark.parametrize("tag", CLIP_TAGS)
def test_numpy_and_c_oracle_match_reference(tag, tmp_path):
    g = load_golden(tag)
    clip = rebuild_clip(g, tmp_path)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n) for n in CAMERA_NAMES]
    for c in cams:
        assert np.array_equal(c["K"], g[f"cal_{c['name']}_K"])
        assert np.array_equal(c["chassis2camera"], g[f"cal_{c['name']}_chassis2camera"])
        assert [c["W"], c["H"], c["W0"], c["H0"]] == g[f"cal_{c['name']}_wh"].tolist()
    maps = _static_maps(clip)
    assert sorted(maps) == sorted(str(d) for d in g["datasets"])
    for ds, static in maps.items():
        assert_instances_equal(static, golden_instances(g, f"{ds}_static"))
        assert str(g[f"{ds}_static_dtype"]) == "float32"
        xyz, col, counts, classes = O.flatten_instances(static)
        stamps, poses = O.pose_track(clip, att, DEFAULT_CAMA_CONFIGS, ds)
        assert np.array_equal(np.asarray(poses), g[f"{ds}_pose_abs"])
        assert np.array_equal(stamps, g[f"{ds}_pose_stamps"])
        seen = []
        for idx, w2c, cropped in O.iter_frames(clip, att, DEFAULT_CAMA_CONFIGS, static, ds):
            seen.append(idx)
            key = f"{ds}_f{idx}"
            assert w2c.dtype == np.float32 and np.array_equal(w2c, g[key + "_w2c"])
            assert_instances_equal(cropped, golden_instances(g, key + "_crop"))
            maps_2d = O.project_all(cropped, cams)
            flat = O.frame_project_flat(xyz, w2c, cams, 960, 540, want_chassis=True)
            # C (FMA chain) crop == reference crop
            gold_crop = g[key + "_crop_points"]
            assert np.array_equal(flat["chassis"][flat["crop_mask"].astype(bool)], gold_crop)
            for ci, c in enumerate(cams):
                gl = golden_instances(g, f"{key}_{c['name']}_vu")
                assert_instances_equal(maps_2d[c["name"]], gl)
                gold_vu = np.concatenate([p for _, p in gl]).reshape(-1, 2) if gl else np.zeros((0, 2))
                vis = flat["vis"][ci].astype(bool)
                # single-point instances go through BLAS gemv in the reference (<= 1 ulp apart): none in fixtures
                assert np.array_equal(flat["vu"][ci][vis], gold_vu)
                # circle stream == truncation + colour + order of the visible list
                circ = g[f"{key}_{c['name']}_circles"]
                ids = np.flatnonzero(vis)
                assert circ.shape[0] == ids.size
                tr = flat["vu"][ci][vis].astype(np.int32)
                assert np.array_equal(circ[:, 0], tr[:, 1]) and np.array_equal(circ[:, 1], tr[:, 0])
                pal = np.asarray([O.GREY_RGB[::-1], O.GOLD_RGB[::-1]], np.int32)
                assert np.array_equal(circ[:, 2:5], pal[col[ids]])
        assert seen == g[f"{ds}_frame_ids"].tolist()


def test_pose_seek_edge_cases():
    g = np.load(join(GOLDEN, "pose_seek.npz"))
    stamps, poses = O.tum_to_poses(g["tum"])
    assert np.array_equal(np.asarray(poses), g["abs_loaded"])
    assert np.array_equal(O.inv_rigid(g["ext"]), g["invT_ext"])
    assert np.array_equal(np.asarray(O.normalize_to_center(poses)), g["abs_normalize2center"])
    rr = O.right_compose(poses, g["ext"])
    assert np.array_equal(np.asarray(rr), g["abs_right_rotate"])
    assert np.array_equal(O.slerp_transform(poses[1], poses[2], 0.3), g["slerp_03"])
    for interp, okk, resk in ((True, "seek_ok", "seek_result"), (False, "seek_nearest_ok", "seek_nearest_result")):
        for q, ok, res in zip(g["queries"], g[okk], g[resk]):
            if ok:
                assert np.array_equal(O.seek_pose(stamps, rr, float(q), 0.5, interp), res)
            else:
                with pytest.raises(RuntimeError):
                    O.seek_pose(stamps, rr, float(q), 0.5, interp)


def test_mosaic_layout():
    g = np.load(join(GOLDEN, "mosaic.npz"))
    imgs = {n: g["img_" + n] for n in CAMERA_NAMES}
    assert np.array_equal(O.mosaic(imgs), g["mosaic"])


def test_circle_footprint_r2_and_clipping():
    # PARITY UNPINNED (no OpenCV on either box): this pins the restatement to the documented
    # 13-pixel diamond of OpenCV's integer midpoint fill for r=2 and to its border clipping.
    img = np.zeros((7, 7, 3), np.uint8)
    O.lib().oracle_circle_fill(img.ctypes.data, 7, 7, img.strides[0], 3, 3, 2, 9, 9, 9)
    want = np.array([[0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0], [0, 0, 1, 1, 1, 0, 0], [0, 1, 1, 1, 1, 1, 0],
                     [0, 0, 1, 1, 1, 0, 0], [0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]], bool)
    assert np.array_equal(img[:, :, 0] == 9, want)
    assert O.circle_halfwidths(2).tolist() == [2, 1, 0]
    assert O.circle_halfwidths(1).tolist() == [1, 0]
    assert O.circle_halfwidths(3).tolist() == [3, 2, 2, 0]   # widths 1,5,5,7,5,5,1: OpenCV pole nubs
    # corners: clipped, never out of bounds
    for (cx, cy) in [(0, 0), (6, 0), (0, 6), (6, 6), (1, 5)]:
        img = np.zeros((7, 7, 3), np.uint8)
        O.lib().oracle_circle_fill(img.ctypes.data, 7, 7, img.strides[0], cx, cy, 2, 1, 2, 3)
        ys, xs = np.nonzero(img[:, :, 0])
        for y, x in zip(ys, xs):
            assert abs(x - cx) <= [2, 1, 0][abs(y - cy)]
        n_expected = sum(1 for dy in range(-2, 3) for dx in range(-2, 3)
                         if abs(dx) <= [2, 1, 0][abs(dy)] and 0 <= cx + dx < 7 and 0 <= cy + dy < 7)
        assert len(ys) == n_expected


def test_render_flat_equals_per_point_python_loop():
    rng = np.random.default_rng(5)
    H, W, N, C = 40, 64, 300, 6
    vu = np.stack([rng.uniform(0, H, (C, N)), rng.uniform(0, W, (C, N))], axis=-1)
    vis = (rng.random((C, N)) < 0.6).astype(np.uint8)
    col = (rng.random(N) < 0.5).astype(np.uint8)
    src = rng.integers(0, 256, (C, H, W, 3), dtype=np.uint8)
    got = O.frame_render_flat(src, vu, vis, col)
    imgs = {}
    for c, name in enumerate(CAMERA_NAMES):
        img = src[c].copy()
        for colour, cls in ((0, "lane_marking"), (1, "Road_teeth")):
            pass
        # one instance per point keeps the draw order == point order
        maps_2d = [{"class": "lane_marking" if col[i] == 0 else "Road_teeth", "points": vu[c, i:i + 1]}
                   for i in range(N) if vis[c, i]]
        imgs[name] = O.render_instances(img, maps_2d)
    assert np.array_equal(got, O.mosaic(imgs))


def test_single_point_instances_measured_deviation(tmp_path, capsys):
    """clip_f_single holds 2-vertex labels shorter than 0.2 units: each densifies to ONE point, which the reference
    pushes through BLAS gemv (4x4 @ 4x1, 3x3 @ 3x1) instead of gemm.  The numpy port of the oracle takes the same
    route (bit-exact, asserted below); the flat C projector --
    the arithmetic the HIP kernels state, a k-ordered FMA chain -- may differ in the last place for exactly those
    instances.  This test MEASURES that: bar 1e-4 px (BASELINE.json), no truncated pixel may flip."""
    from tests.helpers import load_golden, rebuild_clip, single_point_deviation
    from cama_amd.synth import CAMERA_NAMES, DEFAULT_CAMA_CONFIGS
    g = load_golden("f_single")
    clip = rebuild_clip(g, tmp_path)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n) for n in CAMERA_NAMES]
    import json as _json
    labels = _json.load(open(os.path.join(clip, "maps", "map_labels.json")))
    bev = np.load(os.path.join(clip, "maps", "vision_road_mlp_ft.npy"))
    statics = {"cama": O.static_map_cama(bev, labels),
               "nuscenes": O.static_map_nuscenes(_json.load(open(os.path.join(clip, "maps", "map_nuscenes.json"))))}
    for ds, static in statics.items():
        n_one = sum(1 for ins in static if len(ins["points"]) == 1)
        assert n_one >= 30, (ds, n_one)                       # the fixture really contains one-point instances
        xyz, col, counts, classes = O.flatten_instances(static)
        frames = {idx: w2c for idx, w2c, _ in O.iter_frames(clip, att, DEFAULT_CAMA_CONFIGS, static, ds)}

        def project(idx):
            flat = O.frame_project_flat(xyz, frames[idx], cams, cams[0]["W"], cams[0]["H"])
            return [flat["vu"][c][flat["vis"][c].astype(bool)] for c in range(len(cams))]
        # the numpy port takes the reference's own route (per-instance matmul -> gemv for one point): exact
        assert_instances_equal(static, golden_instances(g, f"{ds}_static"))
        for idx, w2c, cropped in O.iter_frames(clip, att, DEFAULT_CAMA_CONFIGS, static, ds):
            assert_instances_equal(cropped, golden_instances(g, f"{ds}_f{idx}_crop"))
            maps_2d = O.project_all(cropped, cams)
            for c in cams:
                assert_instances_equal(maps_2d[c["name"]], golden_instances(g, f"{ds}_f{idx}_{c['name']}_vu"))
        dev_multi, dev_single, n_single, flips = single_point_deviation(g, ds, g[f"{ds}_frame_ids"].tolist(), project)
        with capsys.disabled():
            print(f"\n[f_single/{ds}] FMA-chain projector vs reference: multi-point instances max dev {dev_multi:.3e} px, "
                  f"one-point instances max dev {dev_single:.3e} px over {n_single} projections, {flips} pixel flips")
        assert dev_multi == 0.0
        assert n_single >= 30 and dev_single <= 1e-9 and flips == 0


def _wu_python(src, vu, vis, colour_id, link, radius=2, cols=3):
    """Pure-Python twin of oracle_render_frame_wu (oracle/cama_oracle.c): the same claims and the same blend, written with
    Python integers and floor division -- guards the C restatement against overflow / truncation slips."""
    C, H, W = src.shape[:3]
    pal = np.asarray([O.GREY_RGB[::-1], O.GOLD_RGB[::-1]], np.int64)
    hw = O.circle_halfwidths(radius)
    rows = (C + cols - 1) // cols
    out = np.zeros((rows * H, cols * W, 3), np.uint8)
    for c in range(C):
        claim = {}

        def put(x, y, key1, cov):
            if 0 <= x < W and 0 <= y < H and cov and claim.get((y, x), (0, 0)) < (key1, cov):
                claim[(y, x)] = (key1, cov)
        for i in range(vis.shape[1]):
            if not vis[c, i]:
                continue
            vi, ui = int(vu[c, i, 0]), int(vu[c, i, 1])
            if i > 0 and link[i] and vis[c, i - 1]:
                vp, up = int(vu[c, i - 1, 0]), int(vu[c, i - 1, 1])
                if (vp, up) != (vi, ui):
                    steep = abs(vi - vp) > abs(ui - up)
                    a0, b0, a1, b1 = (vp, up, vi, ui) if steep else (up, vp, ui, vi)
                    if a0 > a1:
                        a0, b0, a1, b1 = a1, b1, a0, b0
                    grad = ((b1 - b0) * 65536) // (a1 - a0)
                    for j in range(a1 - a0 + 1):
                        y = b0 * 65536 + grad * j
                        row, f = y >> 16, (y & 0xffff) >> 8
                        for r, cov in ((row, 255 - f), (row + 1, f)):
                            put(*((r, a0 + j) if steep else (a0 + j, r)), i + 1, cov)
            for dy in range(-radius, radius + 1):
                for dx in range(-hw[abs(dy)], hw[abs(dy)] + 1):
                    put(ui + dx, vi + dy, i + 1, 255)
        img = src[c].astype(np.int64).copy()
        for (y, x), (key1, cov) in claim.items():
            a = cov + (cov >> 7)
            img[y, x] = (pal[int(colour_id[key1 - 1]) & 1] * a + img[y, x] * (256 - a) + 128) >> 8
        r0, q0 = divmod(c, cols)
        out[r0 * H:(r0 + 1) * H, q0 * W:(q0 + 1) * W] = img.astype(np.uint8)
    return out


def test_wu_restatement_in_c_equals_its_python_twin():
    """The anti-aliased segment EXTENSION (no reference semantics): the C definition the HIP kernel is checked against,
    against an independent Python twin, on polylines of every octant incl. points outside the image and broken links."""
    rng = np.random.default_rng(5)
    for trial in range(6):
        C, H, W, N = 3, 48, 80, 40
        src = rng.integers(0, 256, (C, H, W, 3), dtype=np.uint8)
        vu = np.zeros((C, N, 2))
        vu[:, :, 0] = rng.uniform(0, H, (C, N))
        vu[:, :, 1] = rng.uniform(0, W, (C, N))
        if trial % 2:                                   # short steps: the common case (neighbouring polyline points)
            vu[:, 1:] = vu[:, :1] + np.cumsum(rng.uniform(-3, 3, (C, N - 1, 2)), axis=1)
            vu[:, :, 0] = np.clip(vu[:, :, 0], 0, H - 0.01)
            vu[:, :, 1] = np.clip(vu[:, :, 1], 0, W - 0.01)
        vis = (rng.random((C, N)) < 0.85).astype(np.uint8)
        col = rng.integers(0, 2, N).astype(np.uint8)
        link = (rng.random(N) < 0.8)
        link[0] = False
        got = O.frame_render_flat_wu(src, vu, vis, col, link)
        want = _wu_python(src, vu, vis, col, link)
        assert np.array_equal(got, want), trial
        plain = O.frame_render_flat(src, vu, vis, col)
        assert not np.array_equal(got, plain)           # the segments show
    # axis-aligned and 45-degree lines have exact rows: full coverage on the line, nothing beside it
    src = np.zeros((1, 16, 16, 3), np.uint8)
    vu = np.array([[[2.0, 2.0], [2.0, 12.0], [12.0, 12.0], [5.0, 5.0]]])
    out = O.frame_render_flat_wu(src, vu, np.ones((1, 4), np.uint8), np.zeros(4, np.uint8), np.array([0, 1, 1, 1], bool), radius=0)
    grey = tuple(int(v) for v in O.GREY_RGB[::-1])
    assert all(tuple(out[2, x]) == grey for x in range(2, 13)) and not out[1, 5].any() and not out[3, 5].any()
    assert all(tuple(out[y, 12]) == grey for y in range(2, 13)) and not out[7, 11].any()
    assert all(tuple(out[k, k]) == grey for k in range(5, 13)) and not out[6, 5].any()
