"""Host-side frame ingest (cama_amd/frames.py): decode, .npy twins, thread-pool prefetch; and the undistort map builder.
CPU only."""
import numpy as np

from cama_amd import frames as FR
from cama_amd.dataset import ClipManager
from cama_amd.synth import DEFAULT_CAMA_CONFIGS, make_clip


def test_clip_frame_source_decodes_in_order_with_prefetch(tmp_path):
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=6, seed=3, n_lines=2, verts_per_line=3, line_len_m=1.0, raster_size=300,
              image_mode="jpg", image_size=(36, 64), origin_size=(36, 64), with_nuscenes=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(36, 64)), clip)
    src = FR.ClipFrameSource(cm.cm_list, device=None, workers=4, prefetch=2)
    assert src.fused is False                                   # output size == sensor size, no distortion
    got = src._decoded([1, 2])
    assert got.shape == (2, 6, 36, 64, 3) and got.dtype == np.uint8
    for k, idx in enumerate((1, 2)):
        for c, cam in enumerate(cm.cm_list):
            assert np.array_equal(got[k, c], FR.read_bgr(cam.get_image_path(idx, True)))
    assert sorted(src._pending) == [3, 4]                       # two frames ahead are already being decoded
    got5 = src._decoded([5])                                    # jump: stale prefetches are dropped, no leak
    assert got5.shape[0] == 1 and all(k >= 5 for k in src._pending)
    assert np.array_equal(got5[0, 2], FR.read_bgr(cm.cm_list[2].get_image_path(5, True)))


def test_read_bgr_npy_twin_and_channel_order(tmp_path):
    from PIL import Image
    img = np.zeros((8, 12, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 10, 120, 250        # B, G, R
    np.save(tmp_path / "a.npy", img)
    assert np.array_equal(FR.read_bgr(str(tmp_path / "a.jpg")), img)          # .jpg missing -> .npy twin
    assert np.array_equal(FR.read_bgr(str(tmp_path / "a.npy")), img)
    Image.fromarray(img[:, :, ::-1]).save(tmp_path / "b.png")                 # lossless, RGB on disk
    assert np.array_equal(FR.read_bgr(str(tmp_path / "b.png")), img)          # comes back as BGR like cv2.imread


def test_undistort_map_identity_scale_and_monotonic():
    K0 = np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])
    Kn = K0.copy()
    Kn[0] *= 960 / 1600
    Kn[1] *= 540 / 900
    mx, my = FR.undistort_rectify_map(K0, [0.0] * 8, Kn, 960, 540)
    assert mx.dtype == np.float32 and mx.shape == (540, 960)
    jj, ii = np.meshgrid(np.arange(960), np.arange(540))
    assert np.allclose(mx, jj / 0.6, atol=2e-4) and np.allclose(my, ii / 0.6, atol=2e-4)     # src = dst / scale
    assert (mx == mx[0:1]).all() and (my == my[:, 0:1]).all()                                # separable
    mxi, myi = FR.undistort_rectify_map(K0, [], K0, 1600, 900)
    # identity calibration: the float64 round trip through K^-1 and K leaves ~1e-13 px, far below the 1/32 px grid
    assert np.allclose(mxi[0], np.arange(1600), atol=1e-4) and np.allclose(myi[:, 0], np.arange(900), atol=1e-4)
    assert np.array_equal(np.rint(mxi[0] * 32), np.arange(1600) * 32.0)
    mxd, _ = FR.undistort_rectify_map(K0, [-0.2, 0.05, 0, 0, 0, 0, 0, 0], Kn, 960, 540)
    assert not (mxd == mxd[0:1]).all()                                                       # distortion breaks separability


def test_undistort_map_bit_identical_to_the_opencv_order_restatement():
    """The product's vectorised builder against the checker's scalar C restatement of OpenCV's published loop
    (oracle_undistort_map: closed-form inverse, row-incremental sums, no skew term): same bits, for plain scaling,
    full rational + tangential + thin-prism distortion, and a K_new with skew and an off-centre principal point."""
    import pytest
    from oracle import cama_oracle as O
    K0 = np.array([[1266.417203, 0.0, 816.2670197], [0.0, 1266.417203, 491.50706579], [0.0, 0.0, 1.0]])
    Kn = K0.copy()
    Kn[0] *= 960 / 1600
    Kn[1] *= 540 / 900
    Ks = np.array([[700.3, 1.7, 333.1], [0.0, 690.9, 201.7], [0.0, 0.0, 1.0]])
    cases = [(K0, [0.0] * 8, Kn, 960, 540), (K0, [], K0, 160, 90),
             (K0, [-0.21, 0.07, 1e-3, -2e-3, 0.01, 0.02, -0.01, 0.003, 1e-3, -1e-3, 2e-3, 5e-4], Kn, 96, 54),
             (K0, [-0.05, 0.01, 0.0, 0.0, 0.0], Ks, 77, 41)]
    for K_origin, dist, K_new, W, H in cases:
        mx, my = FR.undistort_rectify_map(K_origin, dist, K_new, W, H)
        ox, oy = O.undistort_map(K_origin, dist, K_new, W, H)
        assert np.array_equal(mx.view(np.uint32), ox.view(np.uint32))
        assert np.array_equal(my.view(np.uint32), oy.view(np.uint32))
    # u = fx*xd + u0: a skew entry in K_origin is ignored, like OpenCV does
    Kskew = K0.copy()
    Kskew[0, 1] = 3.0
    a, _ = FR.undistort_rectify_map(Kskew, [-0.1, 0.0, 0.0, 0.0], Kn, 64, 36)
    b, _ = FR.undistort_rectify_map(K0, [-0.1, 0.0, 0.0, 0.0], Kn, 64, 36)
    assert np.array_equal(a, b)
    tilt = [0.0] * 12 + [0.01, 0.0]
    with pytest.raises(NotImplementedError):
        FR.undistort_rectify_map(K0, tilt, Kn, 8, 8)
    with pytest.raises(NotImplementedError):
        O.undistort_map(K0, tilt, Kn, 8, 8)


def test_decode_pump_thread_ends_when_its_source_is_dropped():
    """the decode pump used to be a bound method parked in cv.wait() for ever once a consumer abandoned
    a pass, pinning the source and its decoded batches.  The thread now holds a weak reference only and looks at it twice a
    second."""
    import gc
    import threading
    import weakref

    class FakeSource:
        _pump_depth = 1

        def __init__(self):
            self.steps = []

        def _pump_step(self, P, j, key):
            self.steps.append(key)
            return ("decoded", key)

    keys = [(1, 2), (3, 4), (5, 6), (7, 8)]
    P = {"keys": keys, "index": {k: j for j, k in enumerate(keys)}, "ready": {}, "consumed": 0,
         "cv": threading.Condition(), "stop": False, "error": None}
    src = FakeSource()
    t = threading.Thread(target=FR._pump_entry, args=(weakref.ref(src), P), daemon=True)
    t.start()
    with P["cv"]:
        while keys[0] not in P["ready"]:
            P["cv"].wait(timeout=5)
    assert src.steps == [keys[0]] and t.is_alive()              # one batch ahead, then parked
    with P["cv"]:                                                # the consumer takes it: the pump moves on by one
        P["ready"].pop(keys[0])
        P["consumed"] += 1
        P["cv"].notify_all()
    with P["cv"]:
        while keys[1] not in P["ready"]:
            P["cv"].wait(timeout=5)
    assert t.is_alive()
    del src                                                      # the consumer walks away mid-pass
    gc.collect()
    t.join(timeout=3)
    assert not t.is_alive(), "the pump thread must end once its source is gone"
    # and a cancelled plan ends it at once
    src = FakeSource()
    P2 = dict(P, ready={}, consumed=0, cv=threading.Condition(), stop=False)
    t = threading.Thread(target=FR._pump_entry, args=(weakref.ref(src), P2), daemon=True)
    t.start()
    with P2["cv"]:
        P2["stop"] = True
        P2["cv"].notify_all()
    t.join(timeout=3)
    assert not t.is_alive()


def test_separable_undistort_map_is_the_full_map_bit_for_bit():
    """Zero distortion + an axis-aligned new camera matrix (what every nuScenes camera and the reference's CameraManager
    give, cama/reproject.py:232-240): one row and one column of OpenCV's map carry all of it.  The O(W + H) form must equal
    the full O(W H) restatement bit for bit -- it replaces it on every new clip's first frame -- and must decline (None)
    whenever the map is not of that form."""
    from cama_amd import frames as FR
    rng = np.random.default_rng(5)
    K0 = np.array([[1266.417203046554, 0.0, 816.2670197447984], [0.0, 1266.417203046554, 491.50706579294757], [0, 0, 1.0]])
    for H, W in ((540, 960), (450, 800), (97, 161), (900, 1600)):
        for _ in range(2):
            Kn = K0.copy()
            Kn[0, :] *= W / 1600
            Kn[1, :] *= H / 900
            mx, my = FR.undistort_rectify_map(K0, [], Kn, W, H)
            for dist in ([], None, np.zeros(5), np.zeros(14)):
                sep = FR.undistort_rectify_map_separable(K0, dist, Kn, W, H)
                assert sep is not None and sep[0].dtype == np.float32 and sep[0].shape == (W,) and sep[1].shape == (H,)
                assert np.array_equal(mx, np.broadcast_to(sep[0][None, :], (H, W)))
                assert np.array_equal(my, np.broadcast_to(sep[1][:, None], (H, W)))
            K0 = K0 + np.diag([rng.uniform(-40, 40), rng.uniform(-40, 40), 0.0])      # another calibration
            K0[0, 2] += rng.uniform(-9, 9)
    Kn = K0.copy()
    Kn[0, 1] = 0.25                                                    # skew: rows differ
    assert FR.undistort_rectify_map_separable(K0, [], Kn, 960, 540) is None
    Kn = K0.copy()
    Kn[2, 0] = 1e-6                                                    # projective row
    assert FR.undistort_rectify_map_separable(K0, [], Kn, 960, 540) is None
    assert FR.undistort_rectify_map_separable(K0, [0.0, 0.0, 1e-4, 0.0, 0.0], K0, 960, 540) is None    # any distortion

    class CM:                                                          # the attributes camera_maps_compact reads
        K_origin, d, d_origin, width, height = K0, [], np.zeros(5), 960, 540
        K = K0 * np.array([[0.6], [0.6], [1.0]])
    mx, my, sep = FR.camera_maps_compact(CM)
    full = FR.undistort_rectify_map(CM.K_origin, [], CM.K, 960, 540)
    assert sep == 1 and np.array_equal(full[0][0], mx) and np.array_equal(full[1][:, 0], my)

    class CMd:                                                         # (its own class: the maps are cached on the object)
        K_origin, d, width, height, K = CM.K_origin, [], 960, 540, CM.K
        d_origin = np.array([-0.05, 0.01, 0.0, 0.0, 0.0])
    mx, my, sep = FR.camera_maps_compact(CMd)
    assert sep == 0 and mx.shape == (540, 960)


def test_decoded_frame_cache_serves_views_notices_changed_files_and_evicts_oldest_first(tmp_path):
    """ClipFrameSource keeps a clip's decoded frames for its later passes (main.py renders every clip twice from the same
    camera files): whole cached batches come back as views, frames spread over batches as one gather, a file whose size /
    mtime changed drops its batch, and the byte cap evicts the oldest batch first.  (Bookkeeping only: CPU tensors.)"""
    import os
    import torch
    from cama_amd import frames as FR

    class Cam:
        def __init__(self, name):
            self.name = name

        def get_image_path(self, idx, sync):
            return str(tmp_path / f"{self.name}_{idx}.jpg")

        def needs_resample(self):
            return False
    cams = [Cam("a"), Cam("b")]
    for c in cams:
        for i in range(1, 9):
            (tmp_path / f"{c.name}_{i}.jpg").write_bytes(bytes([i]) * (10 + i))
    src = FR.ClipFrameSource(cams, None, decoder="host")
    assert src._cache_cap == 0 and src._cache_lookup([1, 2]) is None            # no device: off
    per_batch = 4 * 2 * 3 * 5 * 3
    src._cache_cap = 2 * per_batch + 1                                           # room for two batches
    b1 = torch.arange(per_batch, dtype=torch.uint8).reshape(4, 2, 3, 5, 3)
    b2 = (b1 + 100)
    src._cache_store([1, 2, 3, 4], b1)
    src._cache_store([5, 6, 7, 8], b2)
    assert src._cache_bytes == 2 * per_batch and src.cache_stats["stored_batches"] == 2
    v = src._cache_lookup([2, 3])
    assert v.data_ptr() == b1[1:3].data_ptr() and torch.equal(v, b1[1:3])       # consecutive rows of one batch: a view
    g = src._cache_lookup([4, 5])                                                # across batches: one gather
    assert torch.equal(g, torch.stack([b1[3], b2[0]])) and src.cache_stats["hits"] == 2
    assert src._cache_lookup([8, 9]) is None                                     # frame 9 was never decoded
    # a camera file of frame 6 changes: its whole batch goes, the other batch is still served
    p = tmp_path / "b_6.jpg"
    p.write_bytes(b"x" * 99)
    assert src._cache_lookup([5, 6]) is None and src.cache_stats["stale"] == 1
    assert src._cache_lookup([7]) is None and src._cache_bytes == per_batch
    assert src._cache_lookup([1, 2, 3, 4]).data_ptr() == b1.data_ptr()
    # the cap: a third batch evicts the OLDEST one
    src._cache_store([5, 6, 7, 8], b2)
    src._cache_lookup([1])                                                       # touch batch 1: batch (5..8) is now the older one
    b3 = b1 + 7
    os.utime(tmp_path / "a_1.jpg")                                               # (same size, new mtime: also a change)
    for c in cams:
        for i in range(9, 13):
            (tmp_path / f"{c.name}_{i}.jpg").write_bytes(b"z" * i)
    src._cache_store([9, 10, 11, 12], b3)
    assert src.cache_stats["evicted_batches"] == 1 and (5, 6, 7, 8) not in src._cache and (9, 10, 11, 12) in src._cache
    assert src._cache_lookup([1, 2]) is None and src.cache_stats["stale"] == 2   # the touched mtime is noticed too
