"""Randomised differential test: fused render (C ABI) vs the C oracle over random rigs / sizes / radii / maps.
CAMA_FUZZ_ITERS overrides the number of cases (default 40)."""
import os

import numpy as np
import pytest

from oracle import cama_oracle as O

pytestmark = pytest.mark.gpu


def _case(rng):
    from scipy.spatial.transform import Rotation
    W = int(rng.choice([640, 960, 640, 160, 333] if os.environ.get("CAMA_FUZZ_WIDE") else [16, 48, 64, 100, 160, 272, 333, 640]))
    H = int(rng.integers(9, 150))
    C = int(rng.integers(1, 10))
    F = int(os.environ.get("CAMA_FUZZ_FRAMES", 0)) or int(rng.integers(1, 4))
    N = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 3000, 5000, 20000]))
    radius = int(rng.choice([0, 1, 2, 2, 2, 3]))
    f64 = bool(rng.random() < 0.3)
    kind = rng.choice(["cloud", "line", "clump", "site"])
    if kind == "site":                     # site-sized: most vertex blocks outside the crop box (block-AABB cull)
        xyz = rng.uniform([-300, -300, -1], [300, 300, 1], (N, 3))
        xyz[: N // 3] = rng.uniform([-30, -40, -1], [40, 40, 1], (N // 3, 3))
        if rng.random() < 0.5:
            xyz = xyz[np.argsort(np.floor(xyz[:, 0] / 20) * 1000 + np.floor(xyz[:, 1] / 20), kind="stable")]
    elif kind == "cloud":
        xyz = rng.uniform([-40, -60, -1], [40, 60, 1], (N, 3))
    elif kind == "line":
        t = np.linspace(0, 1, N)[:, None]
        xyz = np.array([2.0, -3.0, 0.0]) + t * np.array([45.0, 7.0, 0.2]) + rng.normal(0, 0.01, (N, 3))
    else:
        xyz = rng.normal([8.0, 0.0, 1.4], [0.3, 0.3, 0.2], (N, 3))
    xyz = xyz.astype(np.float64 if f64 else np.float32)
    col = (rng.random(N) < 0.5).astype(np.uint8)
    cams = []
    for k in range(C):
        T = np.eye(4)
        yaw = rng.normal(0, 0.15) if k == 0 else rng.uniform(-np.pi, np.pi)      # camera 0 looks down +x, at the map
        T[:3, :3] = Rotation.from_euler("zyx", [yaw, rng.normal(0, 0.05), rng.normal(0, 0.05)]).as_matrix() @ \
            np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]]).T
        T[:3, 3] = rng.normal(0, 1.0, 3) + (np.array([0.0, 0.0, 1.5]) if k == 0 else 0.0)
        K = np.array([[rng.uniform(0.4, 1.2) * W, 0.0, W / 2 + rng.normal(0, 3)],
                      [0.0, rng.uniform(0.4, 1.2) * W, H / 2 + rng.normal(0, 3)], [0.0, 0.0, 1.0]])
        r = rng.random()
        if r < 0.2:
            K[0, 1] = rng.normal(0, 2.0)                      # skew
        elif r < 0.3:
            K[2, 0], K[2, 1] = rng.normal(0, 1e-3, 2)          # projective last row: the general (non-pinhole) path
        elif r < 0.35:
            K[2, 2] = -1.0                                     # flipped depth sign
        cams.append({"name": f"c{k}", "chassis2camera": np.linalg.inv(T), "K": K, "W": W, "H": H})
    w2c = []
    for f in range(F):
        M = np.eye(4)
        M[:3, :3] = Rotation.from_euler("z", rng.uniform(-0.2, 0.2)).as_matrix()
        M[:3, 3] = rng.normal(0, 1.0, 3)
        w2c.append(np.linalg.inv(M.astype(np.float32)))
    crop = [-50, 50, -100, 100, -200, 200] if rng.random() < 0.6 else \
        sorted(rng.uniform(-30, 30, 2).tolist()) + sorted(rng.uniform(-30, 30, 2).tolist()) + [-5.0, 5.0]
    if os.environ.get("CAMA_FUZZ_POSES") == "odd":
        # world->chassis matrices the candidate pre-pass must not be fooled by (it bounds the crop box's pre-image in world
        # space through an inverse it has to verify): any heading and tilt, survey-grid offsets, scale + shear, rank 2,
        # NaN / inf entries, an unbounded crop box
        mode = str(rng.choice(["tilt", "utm", "affine", "singular", "nan", "open_crop"]))
        shift = np.zeros(3)
        if mode == "utm":
            shift = np.array([4.1e5, 5.3e6, 31.0])
            xyz = (xyz.astype(np.float64) + shift).astype(np.float64 if f64 else np.float32)
        w2c = []
        for f in range(F):
            M = np.eye(4)
            M[:3, :3] = Rotation.from_euler("zyx", [rng.uniform(-np.pi, np.pi), rng.normal(0, 0.3), rng.normal(0, 0.3)]).as_matrix()
            M[:3, 3] = rng.normal(0, 5.0, 3) + shift
            Wc = np.linalg.inv(M)
            if mode == "affine":
                Wc[:3, :] = (np.diag(rng.uniform(0.2, 3.0, 3)) + rng.normal(0, 0.2, (3, 3))) @ Wc[:3, :]
            elif mode == "singular" and f == 0:
                Wc[1, :3] = 2.0 * Wc[0, :3]                # rank 2: the whole world maps onto a plane
            elif mode == "nan" and f == 0:
                Wc[int(rng.integers(0, 3)), int(rng.integers(0, 4))] = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
            w2c.append(Wc)
        if mode == "open_crop":
            crop = [-np.inf, 50.0, -100.0, np.inf, -200.0, 200.0]
        kind = f"{kind}/{mode}"
    return dict(W=W, H=H, C=C, F=F, N=N, radius=radius, xyz=xyz, col=col, cams=cams, w2c=np.stack(w2c), crop=crop,
                sort=bool(rng.random() < 0.3), kind=str(kind))


def _fuzz_child(extra_env):
    import subprocess
    import sys
    env = dict(os.environ, **dict({"CAMA_FUZZ_ITERS": os.environ.get("CAMA_FUZZ_ITERS", "40")}, **extra_env))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        __file__ + "::test_fuzz_against_oracle"], env=env, capture_output=True, text=True,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_fuzz_block_index_and_camera_masks():
    """Same fuzz in a child process with CAMA_TEST_HOOKS bounds_min_verts=1: every map hands its block AABBs to the render, so
    k_block_cameras runs for all of these rigs -- skewed K, projective last rows, flipped depth signs, odd crop boxes --
    and the projection skips cameras / blocks by its masks; the output must not change by a byte."""
    _fuzz_child({"CAMA_TEST_HOOKS": "bounds_min_verts=1", "CAMA_FUZZ_SEED": "4242"})


def test_fuzz_work_list_path():
    """... and with cull_list_min=1 on top: every site-sized map additionally goes through the work lists and the
    persistent k_frames_project_list instead of the grid launch."""
    _fuzz_child({"CAMA_TEST_HOOKS": "bounds_min_verts=1,cull_list_min=1", "CAMA_FUZZ_SEED": "77"})


def test_fuzz_candidate_prepass_odd_poses():
    """The work-list path again (candidate pre-pass: world-space crop AABB per frame -> candidate lists -> exact tests) with
    world->chassis matrices that stress the inverse it relies on -- full rotations with tilt, survey-grid translations of
    5e6 m, scale + shear, a rank-2 matrix, NaN / inf entries -- and a crop box open to infinity.  Bit-exact against the
    oracle, i.e. the pre-pass never drops a vertex the per-vertex test keeps."""
    _fuzz_child({"CAMA_TEST_HOOKS": "bounds_min_verts=1,cull_list_min=1", "CAMA_FUZZ_SEED": "505", "CAMA_FUZZ_POSES": "odd",
                 "CAMA_FUZZ_ITERS": "60"})
    # launches of 130 frames: the candidate search runs over three frame chunks (64 + 64 + 2) per box
    _fuzz_child({"CAMA_TEST_HOOKS": "bounds_min_verts=1,cull_list_min=1", "CAMA_FUZZ_SEED": "506", "CAMA_FUZZ_FRAMES": "130",
                 "CAMA_FUZZ_ITERS": "8"})
    # and the same cases through the one-kernel pre-pass it replaced (A/B switch): the oracle agrees with both
    _fuzz_child({"CAMA_TEST_HOOKS": "bounds_min_verts=1,cull_list_min=1,no_candidates", "CAMA_FUZZ_SEED": "505", "CAMA_FUZZ_POSES": "odd",
                 "CAMA_FUZZ_ITERS": "20"})


def test_fuzz_several_vertex_blocks_per_workgroup():
    """... and with project_vb forced: the projection runs 3 (ragged last chunk) / 8 vertex blocks per workgroup as it
    does on launches with >= 16 k (block, frame) items, with the per-wave camera masks and without any."""
    _fuzz_child({"CAMA_TEST_HOOKS": "bounds_min_verts=1,project_vb=3", "CAMA_FUZZ_SEED": "31"})
    _fuzz_child({"CAMA_TEST_HOOKS": "no_bounds,project_vb=8", "CAMA_FUZZ_SEED": "32"})


def test_fuzz_through_the_pipeline_with_both_band_heights():
    """The same fuzz through cama_pipeline_render (pipeline-owned scratch, planned launches for the site-sized maps) -- what the
    product runs -- once with the pipeline's own band height and once with 8-row bands forced (they apply from W >= 600: the
    W = 640 / 960 cases)."""
    _fuzz_child({"CAMA_FUZZ_PIPELINED": "1", "CAMA_FUZZ_WIDE": "1", "CAMA_TEST_HOOKS": "bounds_min_verts=1", "CAMA_FUZZ_SEED": "611"})
    _fuzz_child({"CAMA_FUZZ_PIPELINED": "1", "CAMA_FUZZ_WIDE": "1", "CAMA_TEST_HOOKS": "bounds_min_verts=1,band_rows=8,cull_list_min=1",
                 "CAMA_FUZZ_SEED": "612"})


def test_fuzz_against_oracle():
    import torch
    from cama_amd.engine import Engine
    iters = int(os.environ.get("CAMA_FUZZ_ITERS", "40"))
    rng = np.random.default_rng(int(os.environ.get("CAMA_FUZZ_SEED", "2024")))
    engines = {}
    for it in range(iters):
        c = _case(rng)
        e = engines.setdefault(c["radius"], Engine("cuda:0", radius=c["radius"]))
        rig = e.make_rig([k["name"] for k in c["cams"]], [k["chassis2camera"] for k in c["cams"]],
                         [k["K"] for k in c["cams"]], c["W"], c["H"])
        dmap = e.upload_map(c["xyz"], c["col"], spatial_sort=True if c["sort"] else False)
        src = rng.integers(0, 256, (c["F"], c["C"], c["H"], c["W"], 3), dtype=np.uint8)
        band = e.lib.cama_overlay_band_rows(c["W"])
        if 2 * c["radius"] > band:
            continue
        if os.environ.get("CAMA_FUZZ_PIPELINED"):
            # the product's path: the pipeline (its own demand-sized scratch, planned launches, band height per launch)
            dev_out = torch.full(e.mosaic_shape(rig, c["F"]), 0xA5, dtype=torch.uint8, device="cuda:0")
            e.render_frames_pipelined(dmap, rig, c["w2c"], torch.from_numpy(src).cuda(), dev_out, crop=c["crop"])
            e.join()
            torch.cuda.synchronize()
            out = dev_out.cpu().numpy()
        else:
            out = e.render_frames(dmap, rig, c["w2c"], torch.from_numpy(src).cuda(), crop=c["crop"]).cpu().numpy()
        vu, vis, _ = (t.cpu().numpy() for t in e.project_frames(dmap, rig, c["w2c"], crop=c["crop"]))
        tag = {k: c[k] for k in ("W", "H", "C", "F", "N", "radius", "sort", "kind")}
        for f in range(c["F"]):
            flat = O.frame_project_flat(np.ascontiguousarray(c["xyz"]), c["w2c"][f], c["cams"], c["W"], c["H"], crop=c["crop"])
            assert np.array_equal(vis[f], flat["vis"]), (it, tag)
            m = flat["vis"].astype(bool)
            assert np.array_equal(vu[f][m], flat["vu"][m]), (it, tag)
            want = O.frame_render_flat(src[f], flat["vu"], flat["vis"], c["col"], radius=c["radius"])
            for cam in range(c["C"]):
                r, q = divmod(cam, 3)
                a = out[f, r * c["H"]:(r + 1) * c["H"], q * c["W"]:(q + 1) * c["W"]]
                b = want[r * c["H"]:(r + 1) * c["H"], q * c["W"]:(q + 1) * c["W"]]
                assert np.array_equal(a, b), (it, tag, cam, int((a != b).any(axis=-1).sum()))
