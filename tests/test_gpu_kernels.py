"""GPU parity: the HIP kernels, called through the C ABI (cama_amd.engine -> libcama_hip.so),
against (a) the golden vectors captured from the reference and (b) the C oracle on seeded inputs.
Bar: coordinates, masks and overlay bytes BIT-EXACT (fp64 FMA chains on both sides)."""
import json
from os.path import join

import numpy as np
import pytest

from oracle import cama_oracle as O
from tests.helpers import (CAMERA_NAMES, CLIP_TAGS, DEFAULT_CAMA_CONFIGS, golden_instances, load_golden,
                           rebuild_clip)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from cama_amd.engine import Engine
    return Engine("cuda:0")


def _static_maps(clip):
    maps = {}
    try:
        maps["cama"] = O.static_map_cama(np.load(join(clip, "maps", "vision_road_mlp_ft.npy")),
                                        json.load(open(join(clip, "maps", "map_labels.json"))))
    except FileNotFoundError:
        pass
    try:
        maps["nuscenes"] = O.static_map_nuscenes(json.load(open(join(clip, "maps", "map_nuscenes.json"))))
    except FileNotFoundError:
        pass
    return maps


def _rig(engine, cams):
    return engine.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams],
                           [c["K"] for c in cams], cams[0]["W"], cams[0]["H"])


@pytest.mark.parametrize("tag", CLIP_TAGS)
def test_project_frames_matches_reference_golden(engine, tag, tmp_path):
    g = load_golden(tag)
    clip = rebuild_clip(g, tmp_path)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n) for n in CAMERA_NAMES]
    rig = _rig(engine, cams)
    for ds, static in _static_maps(clip).items():
        xyz, col, counts, classes = O.flatten_instances(static)
        dmap = engine.upload_map(xyz, col)
        frames = [(idx, w2c) for idx, w2c, _ in O.iter_frames(clip, att, DEFAULT_CAMA_CONFIGS, static, ds)]
        assert [i for i, _ in frames] == g[f"{ds}_frame_ids"].tolist()
        w2c = np.stack([m for _, m in frames])
        vu, vis, cm = (t.cpu().numpy() for t in engine.project_frames(dmap, rig, w2c))
        for k, (idx, m) in enumerate(frames):
            key = f"{ds}_f{idx}"
            # crop mask == which points the reference kept (order preserved)
            chassis, cmask = engine.transform_points(xyz, m[None], crop=engine.crop)
            chassis, cmask = chassis.cpu().numpy()[0], cmask.cpu().numpy()[0].astype(bool)
            assert np.array_equal(cm[k].astype(bool), cmask)
            assert np.array_equal(chassis[cmask], g[key + "_crop_points"])
            for ci, name in enumerate(CAMERA_NAMES):
                gl = golden_instances(g, f"{key}_{name}_vu")
                gold = np.concatenate([p for _, p in gl]).reshape(-1, 2) if gl else np.zeros((0, 2))
                got = vu[k, ci][vis[k, ci].astype(bool)]
                assert got.shape == gold.shape, (tag, ds, idx, name, got.shape, gold.shape)
                assert np.array_equal(got, gold), float(np.abs(got - gold).max())   # bit-exact, bar is 1e-4 px
                # the generic (non-fused) API path gives the same answer on the cropped points
                vu2, vis2 = engine.project_points(rig, chassis[cmask])
                got2 = vu2.cpu().numpy()[ci][vis2.cpu().numpy()[ci].astype(bool)]
                assert np.array_equal(got2, gold)


def test_single_point_instances_measured_deviation_on_device(engine, tmp_path, capsys):
    """clip_f_single (one-point instances: the reference's matmuls become BLAS gemv): the kernels' FMA chain is
    bit-exact on every multi-point instance and within 1e-9 px (bar 1e-4) on the one-point ones, no truncated pixel
    flips; the device result equals the C oracle's bit for bit, and the rendered overlay equals the oracle's."""
    import torch
    from tests.helpers import single_point_deviation
    g = load_golden("f_single")
    clip = rebuild_clip(g, tmp_path)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n) for n in CAMERA_NAMES]
    rig = _rig(engine, cams)
    W, H = cams[0]["W"], cams[0]["H"]
    rng = np.random.default_rng(4)
    for ds, static in _static_maps(clip).items():
        xyz, col, counts, classes = O.flatten_instances(static)
        assert int((counts == 1).sum()) >= 30
        dmap = engine.upload_map(xyz, col)
        frames = [(idx, w2c) for idx, w2c, _ in O.iter_frames(clip, att, DEFAULT_CAMA_CONFIGS, static, ds)]
        w2c = np.stack([m for _, m in frames])
        vu, vis, _ = (t.cpu().numpy() for t in engine.project_frames(dmap, rig, w2c))
        pos = {idx: k for k, (idx, _) in enumerate(frames)}

        def project(idx):
            k = pos[idx]
            return [vu[k, c][vis[k, c].astype(bool)] for c in range(len(cams))]
        dev_multi, dev_single, n_single, flips = single_point_deviation(g, ds, [i for i, _ in frames], project)
        with capsys.disabled():
            print(f"\n[f_single/{ds}] device projector vs reference: multi-point max dev {dev_multi:.3e} px, one-point "
                  f"max dev {dev_single:.3e} px over {n_single} projections, {flips} pixel flips")
        assert dev_multi == 0.0 and n_single >= 30 and dev_single <= 1e-9 and flips == 0
        src = rng.integers(0, 256, (len(frames), len(cams), H, W, 3), dtype=np.uint8)
        out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).to(engine.device)).cpu().numpy()
        for k, (idx, m) in enumerate(frames):
            flat = O.frame_project_flat(xyz, m, cams, W, H)
            assert np.array_equal(vis[k], flat["vis"])
            for c in range(len(cams)):
                v = flat["vis"][c].astype(bool)
                assert np.array_equal(vu[k, c][v], flat["vu"][c][v])
            assert np.array_equal(out[k], O.frame_render_flat(src[k], flat["vu"], flat["vis"], col))


def _random_scene(seed, N, F, W, H, C=6, spread=60.0):
    rng = np.random.default_rng(seed)
    xyz = np.stack([rng.uniform(-spread, spread, N), rng.uniform(-spread * 2, spread * 2, N),
                    rng.normal(0, 0.3, N)], axis=-1).astype(np.float32)
    col = (rng.random(N) < 0.4).astype(np.uint8)
    from cama_amd.synth import camera_to_chassis, CAMERA_YAW_DEG, K_NUSCENES_LIKE
    cams = []
    for name in CAMERA_NAMES[:C]:
        T = np.linalg.inv(camera_to_chassis(CAMERA_YAW_DEG[name]) @ _small_rot(rng))
        K = np.array(K_NUSCENES_LIKE)
        K[0] *= W / 1600.0
        K[1] *= H / 900.0
        cams.append({"name": name, "chassis2camera": T, "K": K, "W": W, "H": H})
    w2c = []
    for f in range(F):
        T = np.eye(4)
        a = 0.3 + 0.05 * f
        T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        T[:3, 3] = [3.0 * f, 1.0, 0.02]
        w2c.append(np.linalg.inv(T.astype(np.float32)))
    return xyz, col, cams, np.stack(w2c)


def _small_rot(rng):
    from scipy.spatial.transform import Rotation
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
    return T


@pytest.mark.parametrize("N,F,W,H", [(3000, 2, 960, 540), (20000, 2, 1600, 900), (500, 3, 100, 37),
                                      (0, 1, 64, 32), (257, 1, 48, 21)])
def test_render_frames_byte_identical_to_oracle(engine, N, F, W, H):
    import torch
    xyz, col, cams, w2c = _random_scene(11 + N, max(N, 1), F, W, H)
    xyz, col = xyz[:N], col[:N]
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col)
    rng = np.random.default_rng(99)
    src = rng.integers(0, 256, (F, 6, H, W, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    vu, vis, _ = (t.cpu().numpy() for t in engine.project_frames(dmap, rig, w2c)) if N else (None, None, None)
    stamped = 0
    for f in range(F):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H) if N else \
            {"vu": np.zeros((6, 0, 2)), "vis": np.zeros((6, 0), np.uint8)}
        if N:
            assert np.array_equal(vis[f], flat["vis"])
            m = flat["vis"].astype(bool)
            assert np.array_equal(vu[f][m], flat["vu"][m])
            stamped += int(m.sum())
        want = O.frame_render_flat(src[f], flat["vu"], flat["vis"], col)
        assert out[f].shape == want.shape
        diff = np.flatnonzero((out[f] != want).any(axis=-1))
        assert diff.size == 0, f"{diff.size} differing pixels in frame {f}"
    if N >= 3000:
        assert stamped > N // 10     # the scene really draws something


def test_render_overlap_order_last_writer_wins(engine):
    """Many points of both colours on the same pixels: the per-pixel owner must be the highest draw index."""
    import torch
    W, H, N = 64, 48, 4000
    rng = np.random.default_rng(3)
    # points straight ahead of the front camera, tightly clustered => heavy overlap
    xyz = np.stack([rng.uniform(9.5, 10.5, N), rng.uniform(-0.2, 0.2, N), rng.uniform(1.3, 1.7, N)], -1).astype(np.float32)
    col = (np.arange(N) % 2).astype(np.uint8)
    _, _, cams, _ = _random_scene(5, 10, 1, W, H)
    w2c = np.eye(4, dtype=np.float32)[None]
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col)
    src = rng.integers(0, 256, (1, 6, H, W, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    flat = O.frame_project_flat(xyz, w2c[0], cams, W, H)
    assert flat["vis"].sum() > 1000
    assert np.array_equal(out[0], O.frame_render_flat(src[0], flat["vu"], flat["vis"], col))


def test_float64_vertex_buffer(engine):
    import torch
    xyz, col, cams, w2c = _random_scene(21, 2000, 2, 160, 96)
    xyz64 = xyz.astype(np.float64) + np.random.default_rng(1).normal(0, 1e-9, xyz.shape)   # not fp32-representable
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz64, col)
    assert dmap.is_f64 == 1
    src = np.random.default_rng(2).integers(0, 256, (2, 6, 96, 160, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    for f in range(2):
        flat = O.frame_project_flat(xyz64, w2c[f], cams, 160, 96)
        assert np.array_equal(out[f], O.frame_render_flat(src[f], flat["vu"], flat["vis"], col))


def test_stamp_points_generic_render_maps(engine):
    import torch
    H, W, n = 90, 130, 700
    rng = np.random.default_rng(8)
    vu = np.stack([rng.uniform(0, H, n), rng.uniform(0, W, n)], -1)
    vu[:5] = [[0.2, 0.9], [H - 0.01, W - 0.01], [0.0, W - 1.0], [H - 1.0, 0.0], [1.99, 2.01]]   # borders
    col = (rng.random(n) < 0.5).astype(np.uint8)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    got = engine.stamp_points(torch.from_numpy(img.copy()).cuda(), vu, col).cpu().numpy()
    want = img.copy()
    O.render_instances(want, [{"class": "lane_marking" if c == 0 else "Road_teeth", "points": vu[i:i + 1]}
                              for i, c in enumerate(col)])
    assert np.array_equal(got, want)


def test_radius_variants(engine):
    """Radius is a parameter (table-driven footprint); r = 0, 1, 3, 5, 7 against the oracle's circle (7 = the widest
    footprint of the fused overlay's padded owner table); a larger radius is refused, not mis-drawn."""
    import torch
    from cama_amd import _lib
    from cama_amd.engine import Engine
    with pytest.raises(_lib.CamaHipError):
        e8 = Engine("cuda:0", radius=8)
        xyz, col, cams, w2c = _random_scene(38, 200, 1, 160, 96)
        e8.render_frames(e8.upload_map(xyz, col), _rig(e8, cams), w2c,
                         torch.zeros((1, 6, 96, 160, 3), dtype=torch.uint8, device="cuda"))
    for r in (0, 1, 3, 5, 7):
        e = Engine("cuda:0", radius=r)
        xyz, col, cams, w2c = _random_scene(30 + r, 1500, 1, 160, 96)
        rig = _rig(e, cams)
        dmap = e.upload_map(xyz, col)
        src = np.random.default_rng(r).integers(0, 256, (1, 6, 96, 160, 3), dtype=np.uint8)
        out = e.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
        flat = O.frame_project_flat(xyz, w2c[0], cams, 160, 96)
        assert np.array_equal(out[0], O.frame_render_flat(src[0], flat["vu"], flat["vis"], col, radius=r))


def test_bad_arguments_fail_loudly(engine):
    from cama_amd import _lib
    L = _lib.lib()
    assert L.cama_project_points(None, 10, None, None, 6, 10, 10, None, None, None) == -1
    assert b"NULL" in L.cama_last_error()
    assert L.cama_render_frames(None, None, None, 0, None, None, None, 0, 0, None, 1, None, None, 99, None, 10, 10, None, None, 3, 2,
                                None, None, None, 0, None) == -1
    assert b"C=99" in L.cama_last_error()


def test_disc_table_wider_than_the_radius_is_rejected(engine):
    """The fused overlay pads every LDS owner row by `radius` cells and does not clamp x: a caller-supplied half-width
    table with hw[k] > radius must be refused at the C ABI (CAMA_EINVAL), not rasterised out of bounds."""
    import torch
    from cama_amd import _lib
    xyz, col, cams, w2c = _random_scene(5, 500, 1, 160, 96)
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col)
    src = torch.zeros((1, 6, 96, 160, 3), dtype=torch.uint8, device="cuda")
    want = engine.render_frames(dmap, rig, w2c, src).clone()
    good = engine.halfwidth.copy()
    try:
        engine.halfwidth = np.asarray([2, 3, 0], np.int32)          # hw[1] = 3 > radius 2
        with pytest.raises(_lib.CamaHipError, match="halfwidth"):
            engine.render_frames(dmap, rig, w2c, src)
    finally:
        engine.halfwidth = good
    assert torch.equal(engine.render_frames(dmap, rig, w2c, src), want)


def test_resample_matches_restated_opencv_remap(engine):
    """Undistort+resize kernel vs the oracle's restatement of cv2.initUndistortRectifyMap + cv2.remap."""
    import torch
    from cama_amd import frames as FR
    rng = np.random.default_rng(12)
    H0, W0, H, W = 45, 80, 27, 48          # same 0.6 scale as 900x1600 -> 540x960
    K0 = np.array([[63.3, 0.0, 40.8], [0.0, 63.3, 24.6], [0.0, 0.0, 1.0]])
    Kn = K0.copy()
    Kn[0] *= W / W0
    Kn[1] *= H / H0
    img = rng.integers(0, 256, (3, H0, W0, 3), dtype=np.uint8)
    for dist in ([0.0] * 8, [-0.21, 0.07, 0.001, -0.002, 0.01, 0.0, 0.0, 0.0]):
        mx, my = O.undistort_map(K0, dist, Kn, W, H)
        pmx, pmy = FR.undistort_rectify_map(K0, dist, Kn, W, H)
        assert np.array_equal(mx, pmx) and np.array_equal(my, pmy)      # product map builder == oracle's scalar loops
        if not any(dist):
            jj, ii = np.meshgrid(np.arange(W), np.arange(H))
            assert np.allclose(mx, jj / 0.6, atol=1e-4) and np.allclose(my, ii / 0.6, atol=1e-4)   # src = dst / scale

        class _CM:          # the few attributes Engine.resample reads
            height, width, K_origin, K, d, d_origin = H, W, K0, Kn, [], np.asarray(dist)
        got = engine.resample(_CM(), torch.from_numpy(img).cuda()).cpu().numpy()
        for k in range(3):
            assert np.array_equal(got[k], O.remap_bilinear(img[k], mx, my))
    # out-of-range taps read the constant border 0
    mx2, my2 = mx - 30.0, my + 10.0

    class _CM2:
        height, width, K_origin, K, d, d_origin = H, W, K0, Kn, [], np.zeros(8)
        _resample_maps = (np.ascontiguousarray(mx2), np.ascontiguousarray(my2))
    got = engine.resample(_CM2(), torch.from_numpy(img[0]).cuda()).cpu().numpy()
    assert np.array_equal(got, O.remap_bilinear(img[0], mx2, my2)) and (got == 0).any()


def test_dense_polyline_duplicate_pixels(engine):
    """1 mm-spaced points far from the camera: long runs of consecutive vertices truncate to the same pixel and
    alternate colours; dropping all but the last of each run (done in the binning kernel) must not change a byte."""
    import torch
    W, H, N = 320, 180, 30000
    t = np.linspace(0.0, 1.0, N)
    xyz = np.stack([20.0 + 25.0 * t, -3.0 + 6.0 * t + 0.2 * np.sin(40 * t), 0.02 * np.cos(9 * t)], -1).astype(np.float32)
    col = ((np.arange(N) // 3) % 2).astype(np.uint8)
    _, _, cams, _ = _random_scene(5, 10, 1, W, H)
    w2c = np.stack([np.eye(4, dtype=np.float32), np.linalg.inv(np.array(
        [[1, 0, 0, 2.0], [0, 1, 0, 0.5], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32))])
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col)
    src = np.random.default_rng(4).integers(0, 256, (2, 6, H, W, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    for f in range(2):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
        vis = flat["vis"][1].astype(bool)                      # front camera
        px = flat["vu"][1][vis].astype(np.int32)
        runs = (np.diff(px, axis=0) == 0).all(axis=1).sum()
        assert vis.sum() > 5000 and runs > vis.sum() // 2       # the case really has long duplicate runs
        assert np.array_equal(out[f], O.frame_render_flat(src[f], flat["vu"], flat["vis"], col))


@pytest.mark.parametrize("spatial_sort", [False, True])
def test_interleaved_pixel_runs_keep_the_last_writer(engine, spatial_sort):
    """Round 5: a stamp is dropped in the projection when one of the next 8 lanes of its wave carries a greater draw key on
    the SAME pixel (not just the next lane).  Far-range polylines alternate between two or three pixels (A B A B ..., A B C A
    B C ...); the colours below change with a period that is coprime to those patterns, so dropping the wrong one of two
    stamps on a pixel changes the rendered colour.  Also with a Morton-sorted buffer, where a later lane may carry a SMALLER
    draw key (then nothing may be dropped on its account).  Byte-exact against the oracle."""
    import torch
    W, H = 320, 180
    _, _, cams, _ = _random_scene(5, 10, 1, W, H)
    cam = cams[1]
    Kinv, c2cam_inv = np.linalg.inv(cam["K"]), np.linalg.inv(cam["chassis2camera"])
    rng = np.random.default_rng(11)
    pts = []
    for run in range(60):                                        # runs of 40 .. 200 points hopping between 2 .. 4 pixels
        period = int(rng.integers(2, 5))
        centres = [(float(rng.integers(20, W - 20)) + 0.5, float(rng.integers(20, H - 20)) + 0.5) for _ in range(period)]
        depth = float(rng.uniform(20.0, 60.0))
        for j in range(int(rng.integers(40, 200))):
            u, v = centres[j % period]
            u += float(rng.uniform(-0.3, 0.3))                   # stays inside the pixel
            v += float(rng.uniform(-0.3, 0.3))
            pc = Kinv @ np.array([u * depth, v * depth, depth])
            pts.append((c2cam_inv @ np.r_[pc, 1.0])[:3])
    xyz = np.ascontiguousarray(np.asarray(pts, np.float64))
    N = len(xyz)
    col = ((np.arange(N) // 5) % 2).astype(np.uint8)             # period 10 against hop periods 2, 3, 4
    w2c = np.eye(4, dtype=np.float32)[None]
    crop = [-1e9, 1e9, -1e9, 1e9, -1e9, 1e9]
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col, spatial_sort=spatial_sort)
    assert (dmap.sorted_key is not None) == spatial_sort
    src = np.random.default_rng(4).integers(0, 256, (1, 6, H, W, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda(), crop=crop).cpu().numpy()
    flat = O.frame_project_flat(xyz, w2c[0], cams, W, H, crop=crop)
    vis = flat["vis"][1].astype(bool)
    px = flat["vu"][1][vis].astype(np.int32)
    assert vis.sum() > 4000 and len(np.unique(px[:, 0] * W + px[:, 1])) < 400       # thousands of stamps on a few hundred pixels
    assert np.array_equal(out[0], O.frame_render_flat(src[0], flat["vu"], flat["vis"], col))
    st = engine.bin_stats()
    if not spatial_sort:                 # the waves that show an A B A pattern (period-2 runs) drop theirs in the projection
        assert st["stamps"] < 0.9 * sum(int(flat["vis"][c].sum()) for c in range(6)), st


def test_early_outs_do_not_change_visibility_on_knife_edges(engine):
    """Points placed exactly on / within an ulp of the image borders and of the z = 0 plane of the front camera:
    the bin kernel's depth and frustum early-outs must agree with the full chain (oracle) bit for bit."""
    import torch
    W, H = 160, 96
    _, _, cams, _ = _random_scene(5, 10, 1, W, H)
    cam = cams[1]
    Kinv = np.linalg.inv(cam["K"])
    c2cam_inv = np.linalg.inv(cam["chassis2camera"])
    pts = []
    for u in (0.0, np.nextafter(0.0, -1), 1e-300, -1e-300, W, np.nextafter(float(W), 0), W - 1e-9, W + 0.5, -0.5, W / 2):
        for v in (0.0, np.nextafter(0.0, -1), H, np.nextafter(float(H), 0), H + 0.5, -0.5, H / 2):
            for depth in (1e-6, 0.5, 7.0, 300.0, -1e-6, 0.0, -3.0):
                pc = Kinv @ np.array([u * depth, v * depth, depth])
                pts.append((c2cam_inv @ np.r_[pc, 1.0])[:3])
    xyz = np.asarray(pts, np.float64)
    col = (np.arange(len(xyz)) % 2).astype(np.uint8)
    w2c = np.eye(4, dtype=np.float32)[None]
    crop = [-1e9, 1e9, -1e9, 1e9, -1e9, 1e9]
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col)
    src = np.random.default_rng(4).integers(0, 256, (1, 6, H, W, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda(), crop=crop).cpu().numpy()
    flat = O.frame_project_flat(xyz, w2c[0], cams, W, H, crop=crop)
    vu, vis, _ = (t.cpu().numpy() for t in engine.project_frames(dmap, rig, w2c, crop=crop))
    assert np.array_equal(vis[0], flat["vis"]) and 50 < flat["vis"][1].sum() < len(xyz)
    assert np.array_equal(out[0], O.frame_render_flat(src[0], flat["vu"], flat["vis"], col))


def test_spatially_sorted_map_renders_identically(engine):
    """An unordered (random) vertex buffer is rendered from a Morton-sorted copy keyed by the original draw index:
    bytes must equal the oracle drawing in the ORIGINAL order, with heavy two-colour overlap."""
    import torch
    from cama_amd.engine import lacks_spatial_order, morton_order
    W, H, N = 320, 180, 20000
    rng = np.random.default_rng(17)
    xyz = np.stack([rng.uniform(-60, 60, N), rng.uniform(-110, 110, N), rng.normal(0, 0.2, N)], -1).astype(np.float32)
    xyz[:3000, :2] = rng.normal([12.0, 0.0], 0.4, (3000, 2))          # a dense clump straight ahead: many overlaps
    xyz[:3000] = xyz[rng.permutation(3000)]
    col = (rng.random(N) < 0.5).astype(np.uint8)
    assert lacks_spatial_order(xyz)
    order = morton_order(xyz)
    assert sorted(order.tolist()) == list(range(N))
    _, _, cams, w2c = _random_scene(5, 10, 2, W, H)
    rig = _rig(engine, cams)
    src = rng.integers(0, 256, (2, 6, H, W, 3), dtype=np.uint8)
    auto = engine.upload_map(xyz, col)
    plain = engine.upload_map(xyz, col, spatial_sort=False)
    assert auto.sorted_key is not None and plain.sorted_key is None
    a = engine.render_frames(auto, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    b = engine.render_frames(plain, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(a, b)
    for f in range(2):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
        assert flat["vis"].sum() > 3000
        assert np.array_equal(a[f], O.frame_render_flat(src[f], flat["vu"], flat["vis"], col))
    # API mode is unaffected by the sorted copy (original order)
    vu, vis, _ = (t.cpu().numpy() for t in engine.project_frames(auto, rig, w2c))
    flat = O.frame_project_flat(xyz, w2c[0], cams, W, H)
    assert np.array_equal(vis[0], flat["vis"])


def test_block_bounds_match_numpy(engine):
    """cama_map_bounds: per-wave (cama_map_bounds_block() = 64 vertices) AABBs, ragged tail, NaNs ignored, all-NaN block = empty box."""
    import torch
    from cama_amd import _lib
    L = _lib.lib()
    blk = L.cama_map_bounds_block()
    rng = np.random.default_rng(3)
    for dt in (np.float32, np.float64):
        N = 5 * blk + 37
        xyz = (rng.normal(0, 100, (N, 3))).astype(dt)
        xyz[rng.integers(0, N, 50), rng.integers(0, 3, 50)] = np.nan
        xyz[2 * blk:3 * blk] = np.nan
        dm = engine.upload_map(xyz, np.zeros(N, np.uint8), spatial_sort=False)
        got = dm.bounds.cpu().numpy()
        assert got.shape == (6, 6)
        for b in range(6):
            chunk = xyz[b * blk:(b + 1) * blk].astype(np.float64)
            for k in range(3):
                col = chunk[:, k][~np.isnan(chunk[:, k])]
                lo, hi = (col.min(), col.max()) if col.size else (np.inf, -np.inf)
                assert got[b, 2 * k] == lo and got[b, 2 * k + 1] == hi


@pytest.mark.parametrize("n_l,F", [(120, 5), (600, 15)])
def test_crop_cull_by_block_bounds_is_invisible(engine, n_l, F):
    """Site-sized map (polylines over +-300 m, ~5 % inside the crop box): rendering with the block-AABB cull, without
    it, and the oracle (which crops every vertex, reproject.py:118-131) give the same bytes -- including vertices
    exactly ON the crop faces, blocks straddling them, a frame far outside the map, and NaN vertices.  The second
    size has >= 16384 (block, frame) items: the work-list + persistent-workgroup path (k_cull_blocks)."""
    import torch
    from cama_amd.engine import CROP_BOX
    W, H = 320, 180
    rng = np.random.default_rng(23)
    per = 512
    assert (n_l * per // 256) * F >= 16384 or n_l < 200
    t = (np.arange(per) * 0.1)[None, :]
    p0 = rng.uniform(-300, 250, (n_l, 2))
    ang = rng.uniform(0, 2 * np.pi, n_l)
    xyz = np.stack([p0[:, 0:1] + t * np.cos(ang)[:, None], p0[:, 1:2] + t * np.sin(ang)[:, None],
                    rng.normal(0, 0.05, (n_l, per))], -1).reshape(-1, 3).astype(np.float32)
    # frame 0 is the identity pose: put vertices exactly on each crop face (inclusive bounds) and one ulp outside
    xmin, xmax, ymin, ymax, zmin, zmax = CROP_BOX
    faces = np.array([[xmin, 0, 0], [xmax, 0, 0], [5, ymin, 0], [5, ymax, 0], [5, 1, zmin], [5, 1, zmax]], np.float32)
    outside = faces.copy()
    for k, (axis, sign) in enumerate([(0, -1), (0, 1), (1, -1), (1, 1), (2, -1), (2, 1)]):
        outside[k, axis] = np.nextafter(faces[k, axis], np.float32(sign * np.inf), dtype=np.float32)
    xyz[256 * 3:256 * 3 + 6] = faces
    xyz[256 * 7:256 * 7 + 6] = outside
    xyz[256 * 9 + 5] = np.nan
    col = (rng.random(len(xyz)) < 0.5).astype(np.uint8)
    _, _, cams, _ = _random_scene(5, 10, 1, W, H)
    w2c = [np.eye(4)]
    poses = [(100.0, -50.0, 0.7), (-200.0, 180.0, 2.9), (5000.0, 5000.0, 0.1), (-20.0, 10.0, -1.3)]
    poses += [(rng.uniform(-280, 280), rng.uniform(-280, 280), rng.uniform(-3, 3)) for _ in range(F - 5)]
    for (px, py, a) in poses:
        T = np.eye(4)
        T[:3, :3] = [[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]
        T[:3, 3] = [px, py, 0.01]
        w2c.append(np.linalg.inv(T.astype(np.float32)))
    w2c = np.stack(w2c).astype(np.float64)
    rig = _rig(engine, cams)
    src = rng.integers(0, 256, (F, 6, H, W, 3), dtype=np.uint8)
    with_b = engine.upload_map(xyz, col, spatial_sort=False)
    without = engine.upload_map(xyz, col, spatial_sort=False)
    assert with_b.bounds is not None
    without.bounds = None
    a = engine.render_frames(with_b, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    b = engine.render_frames(without, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    assert np.array_equal(a, b)
    n_in = []
    for f in range(F):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
        n_in.append(int(flat["crop_mask"].sum()))
        assert np.array_equal(a[f], O.frame_render_flat(src[f], flat["vu"], flat["vis"], col))
    assert n_in[3] == 0 and min(n_in[0], n_in[1], n_in[2], n_in[4]) > 50 and max(n_in) < 0.2 * len(xyz), n_in
    # the on-face vertices are inside (inclusive), their one-ulp neighbours are not
    flat0 = O.frame_project_flat(xyz, w2c[0], cams, W, H)
    assert flat0["crop_mask"][256 * 3:256 * 3 + 6].all() and not flat0["crop_mask"][256 * 7:256 * 7 + 6].any()


@pytest.mark.parametrize("C", [1, 4, 9])
def test_other_camera_counts(engine, C):
    """Rigs that are not 2x3: one camera, a ragged last mosaic row (4 = 3 + 1), and more cameras than the bin
    kernel ranks per round (9 > 8)."""
    import torch
    from cama_amd.synth import camera_to_chassis, K_NUSCENES_LIKE
    W, H, N, F = 160, 96, 6000, 2
    rng = np.random.default_rng(40 + C)
    xyz, col, _, w2c = _random_scene(50 + C, N, F, W, H)
    cams = []
    for k in range(C):
        T = np.linalg.inv(camera_to_chassis(360.0 * k / C + 7.0) @ _small_rot(rng))
        K = np.array(K_NUSCENES_LIKE)
        K[0] *= W / 1600.0
        K[1] *= H / 900.0
        cams.append({"name": f"cam{k}", "chassis2camera": T, "K": K, "W": W, "H": H})
    rig = _rig(engine, cams)
    dmap = engine.upload_map(xyz, col, spatial_sort=False)
    src = rng.integers(0, 256, (F, C, H, W, 3), dtype=np.uint8)
    out = engine.render_frames(dmap, rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
    rows = (C + 2) // 3
    assert out.shape == (F, rows * H, 3 * W, 3)
    drawn = 0
    for f in range(F):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
        want = O.frame_render_flat(src[f], flat["vu"], flat["vis"], col)
        # cells of a ragged last row that hold no camera are never written by the kernel: compare camera cells only
        for c in range(C):
            r, q = divmod(c, 3)
            assert np.array_equal(out[f, r * H:(r + 1) * H, q * W:(q + 1) * W], want[r * H:(r + 1) * H, q * W:(q + 1) * W])
        drawn += int(flat["vis"].sum())
    assert drawn > 500


def test_alpha_extension_matches_own_restatement():
    """EXTENSION (the reference is opaque): translucent stamps, checked against the oracle's restatement; alpha = 1
    stays byte-identical to the opaque path."""
    import torch
    from cama_amd.engine import Engine
    for W, H in ((160, 96), (100, 37)):                        # vector path and generic-width path
        xyz, col, cams, w2c = _random_scene(70 + W, 4000, 2, W, H)
        src = np.random.default_rng(W).integers(0, 256, (2, 6, H, W, 3), dtype=np.uint8)
        for alpha256 in (256, 128, 77, 0):
            e = Engine("cuda:0", alpha=alpha256 / 256.0)
            rig = _rig(e, cams)
            out = e.render_frames(e.upload_map(xyz, col), rig, w2c, torch.from_numpy(src).cuda()).cpu().numpy()
            for f in range(2):
                flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
                want = O.frame_render_flat(src[f], flat["vu"], flat["vis"], col, alpha256=alpha256)
                assert np.array_equal(out[f], want), (W, alpha256, f)
                if alpha256 == 0:
                    assert np.array_equal(out[f], O.mosaic({c["name"]: src[f, k] for k, c in enumerate(cams)}))


@pytest.mark.parametrize("chunk_log2", ["0", "3", "31"])
def test_every_workgroup_to_band_mapping_renders_the_same_bytes(chunk_log2, repo_root):
    """The overlay kernels choose per launch between workgroup -> band mappings (round-robin chunks of 32 bands for small
    launches -- what the rest of this suite runs under --, and for big ones whichever of "contiguous per XCD" and the chunks
    the process's own timings favour).  Force the interleaved, another chunked and the contiguous mapping in a subprocess --
    the knob is read once per process -- and run the oracle-parity tests of this file and the raw overlay families under
    each: ragged camera rows, odd widths, several radii, stamped and unstamped bands."""
    import os
    import subprocess
    import sys
    # (round 4's further orders -- stagger, translation look-ahead, XCD groups, band-innermost items -- measured never better and
    # were removed in round 5)
    env = dict(os.environ, CAMA_TEST_HOOKS="overlay_chunk_log2=%s" % chunk_log2)
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "tests/test_gpu_kernels.py", "tests/test_gpu_dropin.py",
                        "-k", "byte_identical_to_oracle or radius_variants or other_camera_counts or last_writer or "
                              "raw_overlay_kernel_families or fused_raw_frame or alpha_extension"],
                       cwd=repo_root, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout


def test_segment_extension_matches_its_own_restatement(engine):
    """EXTENSION (no reference semantics): discs + one-pixel Bresenham segments between neighbouring points
    (cama_stamp_polylines) against the oracle's restatement -- every octant and direction, segments that cross the whole
    image, zero-length segments, two colours overlapping (last writer wins across discs AND segments), unlinked
    neighbours, points on the image border."""
    import torch
    rng = np.random.default_rng(17)
    H, W = 120, 200
    maps_2d = []
    for k in range(40):
        n = int(rng.integers(1, 9))
        pts = np.stack([rng.uniform(0, H - 1e-9, n), rng.uniform(0, W - 1e-9, n)], axis=-1)     # (v, u)
        if k % 5 == 0:
            pts[0] = pts[-1]                                              # a zero-length / closed step
        if k % 7 == 0:
            pts[:, 0] = np.clip(np.round(pts[:, 0] / (H - 1)) * (H - 1), 0, H - 1e-9)   # on the top / bottom rows
        ins = {"class": ["lane_marking", "Road_teeth", "Crosswalk_Line"][k % 3], "points": pts}
        if k % 4 == 0:
            ins["joined"] = np.concatenate([[False], rng.random(n - 1) < 0.5])
        maps_2d.append(ins)
    # long axis-aligned, diagonal and steep segments in all eight octants from one centre
    c = np.array([60.0, 100.0])
    for dv, du in ((0, 90), (0, -90), (55, 0), (-55, 0), (50, 50), (-50, 50), (50, -50), (-50, -50), (20, 90), (55, 30),
                   (-20, -90), (-55, -30), (55, -30), (-20, 90)):
        maps_2d.append({"class": "lane_marking" if dv > 0 else "Stop_Line_x", "points": np.stack([c, c + [dv, du]])})
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    from cama_amd.reproject import colour_id_of, flatten_instances
    vu, counts, classes = flatten_instances(maps_2d, width=2)
    colour = np.repeat(np.asarray([colour_id_of(cl) for cl in classes], np.uint8), counts)
    link = np.concatenate([np.asarray(ins["joined"], bool) if "joined" in ins else np.arange(len(ins["points"])) > 0
                           for ins in maps_2d])
    dev = torch.from_numpy(base.copy()).cuda()
    engine.stamp_points(dev, vu, colour, link=link)
    want = O.render_instances(base.copy(), maps_2d, segments=True)
    got = dev.cpu().numpy()
    assert np.array_equal(got, want)
    plain = O.render_instances(base.copy(), maps_2d)
    assert (want != plain).any(axis=-1).sum() > 500                      # the segments did draw something
    # link = None is the reference's behaviour, untouched
    dev2 = torch.from_numpy(base.copy()).cuda()
    engine.stamp_points(dev2, vu, colour)
    assert np.array_equal(dev2.cpu().numpy(), plain)


def test_antialiased_segments_one_image_at_a_time_match_their_restatement(engine):
    """cama_stamp_polylines_wu (CameraManager.render_maps(..., segments="wu")): the one-image counterpart of the batched Wu
    variant, against the same definition (oracle_render_frame_wu through O.render_instances(segments="wu")) -- every octant,
    zero-length steps, unlinked neighbours, border points, two colours overlapping."""
    import torch
    from cama_amd.reproject import colour_id_of, flatten_instances
    rng = np.random.default_rng(29)
    H, W = 120, 200
    maps_2d = []
    for k in range(40):
        n = int(rng.integers(1, 9))
        pts = np.stack([rng.uniform(0, H - 1e-9, n), rng.uniform(0, W - 1e-9, n)], axis=-1)
        if k % 5 == 0:
            pts[0] = pts[-1]
        if k % 7 == 0:
            pts[:, 0] = np.clip(np.round(pts[:, 0] / (H - 1)) * (H - 1), 0, H - 1e-9)
        ins = {"class": ["lane_marking", "Road_teeth", "Crosswalk_Line"][k % 3], "points": pts}
        if k % 4 == 0:
            ins["joined"] = np.concatenate([[False], rng.random(n - 1) < 0.5])
        maps_2d.append(ins)
    c = np.array([60.0, 100.0])
    for dv, du in ((0, 90), (0, -90), (55, 0), (-55, 0), (50, 50), (-50, 50), (50, -50), (-50, -50), (20, 90), (55, 30),
                   (-20, -90), (-55, -30), (55, -30), (-20, 90)):
        maps_2d.append({"class": "lane_marking" if dv > 0 else "Stop_Line_x", "points": np.stack([c, c + [dv, du]])})
    base = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    vu, counts, classes = flatten_instances(maps_2d, width=2)
    colour = np.repeat(np.asarray([colour_id_of(cl) for cl in classes], np.uint8), counts)
    link = np.concatenate([np.asarray(ins["joined"], bool) if "joined" in ins else np.arange(len(ins["points"])) > 0
                           for ins in maps_2d])
    dev = torch.from_numpy(base.copy()).cuda()
    engine.stamp_points(dev, vu, colour, link=link, wu=True)
    want = O.render_instances(base.copy(), maps_2d, segments="wu")
    got = dev.cpu().numpy()
    if not np.array_equal(got, want):
        bad = np.argwhere((got != want).any(axis=2))
        raise AssertionError(f"{len(bad)} pixels differ, first at {bad[:6].tolist()}")
    hard = O.render_instances(base.copy(), maps_2d, segments=True)
    assert (want != hard).any(axis=-1).sum() > 300                       # partial coverages: another picture than Bresenham's
    # the class surface: CameraManager.render_maps(image, maps, segments="wu") is the same call
    from cama_amd.reproject import CameraManager
    img = base.copy()
    out = CameraManager.render_maps(None, img, maps_2d, segments="wu")
    assert np.array_equal(np.asarray(out), want)


@pytest.mark.parametrize("mode", [True, "wu"])
def test_segment_extension_through_the_class_surface(tmp_path, mode):
    """configs["segments"] = True | "wu": ClipManager.render_vectors draws discs + segments between points that are neighbours
    on the densified polyline AND both visible (no segment across the part of a lane that left the image), image by image;
    equals the oracle's restatement on the materialised maps; the default configs still render the reference's discs."""
    from cama_amd.dataset import ClipManager
    from cama_amd.synth import make_clip
    H, W = 96, 160
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=3, seed=9, n_lines=10, verts_per_line=4, line_len_m=6.0, raster_size=400,
              image_mode="npy", image_size=(H, W), origin_size=(H, W))
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W), segments=mode), clip)
    ref = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W)), clip)
    n_diff = 0
    for (idx, im), (_, im_ref) in zip(cm.yield_frame(dataset="cama"), ref.yield_frame(dataset="cama")):
        maps = cm.project_all_camera(im)
        imgs = cm.render_vectors(maps, idx)
        discs = ref.render_vectors(ref.project_all_camera(im_ref), idx)
        for c in cm.cm_list:
            src = c.read_resized_image_by_index(idx)
            want = O.render_instances(np.ascontiguousarray(src).copy(), maps[c.camera_name], segments=mode)
            assert np.array_equal(np.asarray(imgs[c.camera_name]), want), (idx, c.camera_name, mode)
            assert all("joined" in ins for ins in maps[c.camera_name])
            n_diff += int((np.asarray(discs[c.camera_name]) != want).any(axis=-1).sum())
    assert n_diff >= 0


def test_xcd_dispatch_probe(engine):
    """cama_probe_xcd_map: every block reports an XCD in 0..7; on MI355X consecutive blocks go round-robin over the eight
    (what the XCD-contiguous overlay mapping assumes for speed -- its output does not depend on it)."""
    periodic, first8, per_xcd = engine.xcd_map(4096)
    assert sum(per_xcd) == 4096 and len(per_xcd) == 8 and all(0 <= v < 8 for v in first8)
    assert periodic, (first8, per_xcd)


def _polyline_scene(seed, W, H, F, n_dense=24, n_sparse=10, n_single=5):
    """Polylines for the segment extension: dense ones (1 cm steps, hundreds of points: links across waves and vertex blocks),
    sparse ones (metre-long steps close to the car: segments that cross dozens of rows and several bands, end points that
    leave the image or the crop box) and one-point instances; instance boundaries fall anywhere relative to the 64-vertex
    waves.  Returns xyz (N,3) f32, colour bit0, link (N,) bool, cams, w2c."""
    rng = np.random.default_rng(seed)
    _, _, cams, w2c = _random_scene(seed, 8, F, W, H)
    parts, cols, links = [], [], []
    for k in range(n_dense + n_sparse + n_single):
        if k < n_dense:
            n, step = int(rng.integers(100, 700)), 0.01
        elif k < n_dense + n_sparse:
            n, step = int(rng.integers(2, 12)), float(rng.uniform(0.5, 6.0))
        else:
            n, step = 1, 0.0
        p0 = np.array([rng.uniform(-30, 40), rng.uniform(-40, 40), rng.normal(0, 0.2)])
        ang = rng.uniform(0, 2 * np.pi)
        t = np.arange(n)[:, None] * step
        pts = p0[None] + t * np.array([np.cos(ang), np.sin(ang), 0.0])[None] + rng.normal(0, 0.002, (n, 3)) * (step > 0.1)
        parts.append(pts.astype(np.float32))
        cols.append(np.full(n, k % 2, np.uint8))
        links.append(np.arange(n) > 0)
    return np.concatenate(parts), np.concatenate(cols), np.concatenate(links), cams, w2c


@pytest.mark.parametrize("W,H,F", [(320, 180, 5), (1600, 900, 2)])
def test_antialiased_segments_in_the_fused_path_equal_their_restatement(engine, W, H, F):
    """The Wu variant of the segment extension (CAMA_BIN_SEGMENTS_WU): claims of ((key + 1) << 8 | coverage) in the overlay's
    LDS owner table, records binned one row further at either end, every owned pixel blended once with its own coverage --
    byte-equal to oracle_render_frame_wu on dense and sparse polylines (segments across dozens of rows and several bands, every
    octant, end points out of view), single stream and several pipelined launches in flight."""
    import torch
    xyz, col, link, cams, w2c = _polyline_scene(23 + W, W, H, F)
    rig = engine.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
    dmap = engine.upload_map(xyz, col | (link.astype(np.uint8) << 1), spatial_sort=False)
    src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    got = engine.render_frames(dmap, rig, w2c, src, segments="wu")
    hard = engine.render_frames(dmap, rig, w2c, src, segments=True)
    torch.cuda.synchronize()
    host = src.cpu().numpy()
    for f in range(F):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
        want = O.frame_render_flat_wu(host[f], flat["vu"], flat["vis"], col, link)
        g = got[f].cpu().numpy()
        if not np.array_equal(g, want):
            bad = np.argwhere((g != want).any(axis=2))
            raise AssertionError(f"frame {f}: {len(bad)} pixels differ from the restatement, first at {bad[:5].tolist()}")
        assert (got[f] != hard[f]).any()
    outs = [torch.zeros_like(got) for _ in range(3)]
    for o in outs:
        engine.render_frames_pipelined(dmap, rig, np.asarray(w2c, np.float32), src, o, segments="wu")
    engine.join()
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, got)


@pytest.mark.parametrize("W,H,F", [(320, 180, 5), (1600, 900, 2)])
def test_segment_extension_in_the_fused_path_equals_its_restatement(engine, W, H, F):
    """the north-star's "line segments" in the BATCHED path (CAMA_BIN_SEGMENTS) -- 16-byte records binned to
    every band their segment crosses, rasterised in the overlay's LDS owner table next to the discs -- byte-equal to the
    oracle's restatement (oracle_line_bresenham + the disc, in draw order), for dense and sparse polylines, links across
    wave and block boundaries, end points that drop out of view.  EXTENSION: no reference semantics."""
    import torch
    xyz, col, link, cams, w2c = _polyline_scene(11 + W, W, H, F)
    rig = engine.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
    dmap = engine.upload_map(xyz, col | (link.astype(np.uint8) << 1), spatial_sort=False)
    assert dmap.has_links
    src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    got = engine.render_frames(dmap, rig, w2c, src, segments=True)
    plain = engine.render_frames(dmap, rig, w2c, src)
    torch.cuda.synchronize()
    host = src.cpu().numpy()
    drew = 0
    for f in range(F):
        flat = O.frame_project_flat(xyz, w2c[f], cams, W, H)
        want = O.frame_render_flat_segments(host[f], flat["vu"], flat["vis"], col, link)
        assert np.array_equal(got[f].cpu().numpy(), want), f"frame {f}: segments differ from the restatement"
        # and the disc-only render of the same map is untouched by the link bits
        assert np.array_equal(plain[f].cpu().numpy(), O.frame_render_flat(host[f], flat["vu"], flat["vis"], col)), f
        drew += int((got[f] != plain[f]).any())
    assert drew == F                                          # the segments did change pixels in every frame
    # the same through the pipeline, several launches in flight (the sorted list is re-sized per launch)
    outs = [torch.zeros_like(got) for _ in range(3)]
    for o in outs:
        engine.render_frames_pipelined(dmap, rig, np.asarray(w2c, np.float32), src, o, segments=True)
    engine.join()
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, got)
    # a spatially sorted copy has no polyline neighbours: refused, not drawn wrong
    sorted_map = engine.upload_map(xyz, col | (link.astype(np.uint8) << 1), spatial_sort=True)
    with pytest.raises(Exception):
        engine.render_frames(sorted_map, rig, w2c, src, segments=True)
