"""Product host logic (cama_amd: static-map build, calibration, pose track, frame poses) against the golden
vectors captured from the reference.  CPU only: nothing here touches the GPU or the oracle."""
import json
import os
from os.path import join

import numpy as np
import pytest

from cama_amd.dataset import ClipManager
from cama_amd.dataset_reader import DatasetReader
from cama_amd.pose_transformer import PoseTransformer, SlerpTransform, invT
from cama_amd.reproject import MapManager
from cama_amd.tools import VideoGenerator
from tests.helpers import (CAMERA_NAMES, CLIP_TAGS, DEFAULT_CAMA_CONFIGS, GOLDEN, assert_instances_equal,
                           golden_instances, load_golden, rebuild_clip)


@pytest.mark.parametrize("tag", CLIP_TAGS + ["f_single"])
def test_clip_setup_and_frame_poses(tag, tmp_path):
    g = load_golden(tag)
    clip = rebuild_clip(g, tmp_path)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    assert sorted(cm.instance_maps) == sorted(str(d) for d in g["datasets"])
    for c in cm.cm_list:
        assert np.array_equal(c.K, g[f"cal_{c.camera_name}_K"])
        assert np.array_equal(c.K_origin, g[f"cal_{c.camera_name}_K_origin"])
        assert np.array_equal(c.chassis2camera, g[f"cal_{c.camera_name}_chassis2camera"])
        assert [c.width, c.height, c.width_origin, c.height_origin] == g[f"cal_{c.camera_name}_wh"].tolist()
    for ds in cm.instance_maps:
        assert_instances_equal(cm.instance_maps[ds], golden_instances(g, f"{ds}_static"))
        assert all(ins["points"].dtype == np.float32 for ins in cm.instance_maps[ds])
        dr = DatasetReader(clip)
        pt = cm.get_pt_cama(dr) if ds == "cama" else cm.get_pt_nuscenes(dr)
        assert np.array_equal(np.asarray(pt.absolute_transform), g[f"{ds}_pose_abs"])
        assert np.array_equal(np.asarray(pt.timestamps), g[f"{ds}_pose_stamps"])
        assert np.array_equal(np.asarray(dr.get_sensor_timestamp("camera_front")), g[f"{ds}_frame_stamps"])
        idx, w2c = cm.frame_poses(ds)
        assert idx.tolist() == g[f"{ds}_frame_ids"].tolist()          # skipped frames are skipped
        assert w2c.dtype == np.float32
        for k, i in enumerate(idx):
            assert np.array_equal(w2c[k], g[f"{ds}_f{i}_w2c"])
        # the generator yields the same frames, lazily, without touching the GPU
        frames = list(cm.yield_frame(ds))
        assert [i for i, _ in frames] == idx.tolist()
        assert all(fm._items is None for _, fm in frames)


def test_pose_transformer_surface():
    g = np.load(join(GOLDEN, "pose_seek.npz"))
    pt = PoseTransformer()
    pt.loadarray(g["tum"])
    assert np.array_equal(np.asarray(pt.absolute_transform), g["abs_loaded"])
    assert np.array_equal(np.asarray(pt.relative_transform), g["rel_loaded"])
    pt.right_rotate(g["ext"])
    assert np.array_equal(np.asarray(pt.absolute_transform), g["abs_right_rotate"])
    p2 = PoseTransformer()
    p2.loadarray(g["tum"])
    p2.normalize2center()
    assert np.array_equal(np.asarray(p2.absolute_transform), g["abs_normalize2center"])
    assert np.array_equal(invT(g["ext"]), g["invT_ext"])
    assert np.array_equal(SlerpTransform(g["abs_loaded"][1], g["abs_loaded"][2], 0.3), g["slerp_03"])
    for interp, okk, resk in ((True, "seek_ok", "seek_result"), (False, "seek_nearest_ok", "seek_nearest_result")):
        for q, ok, res in zip(g["queries"], g[okk], g[resk]):
            if ok:
                assert np.array_equal(pt.seek_by_timestamp(float(q), 0.5, interp), res)
            else:
                with pytest.raises(RuntimeError):
                    pt.seek_by_timestamp(float(q), 0.5, interp)
    ok, T = pt.seek_many(g["queries"], 0.5)
    assert ok.tolist() == g["seek_ok"].astype(bool).tolist()
    assert np.array_equal(T[ok], g["seek_result"][ok])          # batched == scalar == reference, bit for bit
    with pytest.raises(AssertionError):
        pt.seek_by_timestamp(1, 0.5, True)                         # ints are rejected like the reference
    # round trips of the wider surface
    tum = pt.dumparray()
    p3 = PoseTransformer()
    p3.loadarray(tum)
    assert np.allclose(np.asarray(p3.absolute_transform), np.asarray(pt.absolute_transform), atol=1e-12)
    assert pt.as_euler(absolute=True).shape == (9, 3) and pt.as_axis_angle(absolute=False).shape == (8, 3)
    assert pt.as_translations(absolute=True).shape == (9, 3) and pt.as_transform().shape == (9, 4, 4)
    p3.normalize2origin()
    assert np.allclose(p3.absolute_transform[0], np.eye(4), atol=1e-12)
    kitti = np.asarray(pt.absolute_transform)[:, :3, :].reshape(-1, 12)
    p4 = PoseTransformer()
    p4.loadarray(kitti, style="kitti")
    assert np.array_equal(np.asarray(p4.absolute_transform), np.asarray(pt.absolute_transform))
    with pytest.raises(NotImplementedError):
        p4.loadarray(kitti, style="nope")
    empty = PoseTransformer()
    with pytest.raises(RuntimeError):
        empty.seek_by_timestamp(1.0, 0.5)


def test_dataset_reader_graph_and_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        DatasetReader(str(tmp_path / "nope"))
    rng = np.random.default_rng(0)
    from scipy.spatial.transform import Rotation

    def rigid():
        T = np.eye(4)
        T[:3, :3] = Rotation.from_rotvec(rng.normal(0, 0.5, 3)).as_matrix()
        T[:3, 3] = rng.normal(0, 1, 3)
        return T
    a2b, c2b, c2d = rigid(), rigid(), rigid()
    att = {"calibration": {"a_2_b": a2b.tolist(), "c_2_b": c2b.tolist(), "c_2_d": c2d.tolist(),
                           "cam": {"K": np.eye(3).tolist(), "d": [0.0] * 8, "image_width": 8, "image_height": 4, "fov": 1}},
           "sync": {"cam": [1000, 1500]}, "unsync": {"cam": [1000, 1250, 1500]}}
    (tmp_path / "p").mkdir()
    json.dump(att, open(tmp_path / "p" / "attribute.json", "w"))
    dr = DatasetReader(str(tmp_path / "p"))
    assert np.array_equal(dr.get_extrinsic("a", "b"), a2b)
    assert np.array_equal(dr.get_extrinsic("b", "a"), invT(a2b))
    assert dr.get_extrinsic("a", "a").dtype == np.float32
    assert dr.get_extrinsic_path("a", "d") == ["a", "b", "c", "d"]
    want = c2d @ (invT(c2b) @ (a2b @ np.eye(4, dtype=np.float32)))
    assert np.array_equal(dr.get_extrinsic("a", "d"), want)
    assert dr.get_extrinsic("a", "zzz") is None
    assert dr.get_sensor_timestamp("cam") == [1.0, 1.5] and dr.get_sensor_timestamp("cam", sync=False)[1] == 1.25
    assert sorted(dr.get_all_sensors()) == ["a", "b", "c", "cam", "d"]
    assert dr.get_intrinsics("cam")["width"] == 8
    assert [p.split("/")[-1] for p in dr.yield_sensor_filepath("cam", "jpg")] == ["1000.jpg", "1500.jpg"]


def test_densify_edge_cases():
    mm = MapManager()
    # one vertex / empty: skipped; a label whose segments all round to num == 0 crashes the reference with IndexError
    assert mm.load_3d_instance_maps([{"attrs": {"type": "x"}, "data": [[0, 0]]}, {"attrs": {"type": "x"}, "data": []}]) == []
    with pytest.raises(IndexError):
        mm.load_3d_instance_maps([{"attrs": {"type": "x"}, "data": [[1.0, 1.0], [1.02, 1.03]]}])
    with pytest.raises(IndexError):
        mm.calculate_3d_instance_maps(np.zeros((4, 4), np.float32), [{"attrs": {"type": "x"}, "data": [[1.0, 1.0], [1.02, 1.03]]}])
    out = mm.load_3d_instance_maps([{"attrs": {"type": "lane_marking"}, "data": [[0, 0], [0.35, 0]]}])
    assert out[0]["points"].shape == (3, 3) and out[0]["points"].dtype == np.float32   # end point never emitted
    # float64 raster -> float64 points, like numpy's concatenate promotion in the reference
    out = mm.calculate_3d_instance_maps(np.zeros((8, 8), np.float64), [{"attrs": {"type": "a"}, "data": [[1, 1], [3, 1]]}])
    assert out[0]["points"].dtype == np.float64


def test_mosaic_layout_plain_dict():
    g = np.load(join(GOLDEN, "mosaic.npz"))
    vg = object.__new__(VideoGenerator)
    assert np.array_equal(vg.concate_image({n: g["img_" + n] for n in CAMERA_NAMES}), g["mosaic"])


def test_library_exports_every_declared_symbol(repo_root):
    """The C-ABI library loads (no GPU needed) and exports exactly what the two headers declare: include/cama_hip.h = the
    contract (every entry listed in INTEGRATION.md), include/cama_hip_diag.h = diagnostics / live timing / options."""
    import re
    import subprocess
    from cama_amd import _lib
    L = _lib.lib()
    header = open(join(repo_root, "include", "cama_hip.h")).read()
    diag = open(join(repo_root, "include", "cama_hip_diag.h")).read()
    fn = r"^(?:int|int64_t|size_t|const char \*)\s*\*?(cama_[a-z_0-9]+)\s*\("
    contract = set(re.findall(fn, header, flags=re.M))
    diagnostic = set(re.findall(fn, diag, flags=re.M))
    assert not contract & diagnostic
    assert diagnostic == set(_lib.DIAG), diagnostic ^ set(_lib.DIAG)
    declared = contract | diagnostic
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(L, name) is not None
    # ... and nothing else: every cama_* symbol the shared object exports is declared
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("cama_")}
    assert exported == declared, exported ^ declared
    # the contract is documented entry by entry
    integ = open(join(repo_root, "INTEGRATION.md")).read()
    missing = sorted(n for n in contract if n not in integ)
    assert not missing, f"INTEGRATION.md does not mention {missing}"
    assert L.cama_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define CAMA_ABI_VERSION (\d+)", header).group(1))
    assert _lib.circle_halfwidths(2).tolist() == [2, 1, 0]
    assert L.cama_render_scratch_bytes(10000, 40, 6, 900, 1600, 2) > 0
    # argument validation happens before any device work
    assert L.cama_project_points(None, 5, None, None, 99, 4, 4, None, None, None) == -1
    assert b"C=99" in L.cama_last_error()


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cama_amd import _lib
    from cama_amd.engine import Engine
    with pytest.raises(_lib.CamaHipError):
        Engine("cuda:0")


def test_dataset_reader_sensor_iterators_match_reference(tmp_path):
    """lidar / IMU / GNSS / wheel iterators and the GNSS / wheel -> TUM converters (dataset_reader.py:45-93,296-407)
    on the synthetic packs of tests/golden/gen_golden.py (current and deprecated log formats), against the
    reference's outputs captured in dataset_reader.npz; exact (same numpy / scipy element-wise arithmetic)."""
    import importlib.util
    import warnings
    from cama_amd.dataset_reader import DatasetReader
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(os.path.dirname(__file__), "golden",
                                                                             "gen_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "dataset_reader.npz"))
    for legacy in (0, 1):
        dr = DatasetReader(gen.make_sensor_pack(str(tmp_path / f"pack{legacy}"), bool(legacy)))
        tag = f"p{legacy}_"
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert np.array_equal(dr.get_GNSS_tum(), G[tag + "gnss_tum"])
            assert np.array_equal(dr.get_wheel_tum(), G[tag + "wheel_tum_unsync"])
            assert np.array_equal(dr.get_wheel_tum(sync=True), G[tag + "wheel_tum_sync"])
            assert len(w) == int(G[tag + "n_warnings"])              # deprecated formats warn once per frame
        for deskewed in (0, 1):
            sweeps = list(dr.yield_lidar(start_idx=1, deskewed=bool(deskewed)))
            assert np.array_equal([t for t, _ in sweeps], G[f"{tag}lidar{deskewed}_t"])
            assert [len(p) for _, p in sweeps] == G[f"{tag}lidar{deskewed}_n"].tolist()
            assert np.array_equal([p.sum() for _, p in sweeps], G[f"{tag}lidar{deskewed}_sum"])
            assert all(p.dtype == np.float64 and p.shape[1] == 6 for _, p in sweeps)
        for name, it in (("imu", dr.yield_IMU()), ("gnss", dr.yield_GNSS()), ("wheel", dr.yield_wheel()),
                         ("wheel_unsync", dr.yield_wheel(sync=False))):
            frames = list(it)
            assert np.array_equal([t for t, _ in frames], G[f"{tag}{name}_t"])
            assert json.dumps([f for _, f in frames], sort_keys=True) == str(G[f"{tag}{name}_json"])


def test_dataset_reader_image_iterators(tmp_path):
    """yield_camera / yield_semantic: Pillow decode, OpenCV channel order (the reference uses cv2.imread, :72-83)."""
    from PIL import Image
    from cama_amd.dataset_reader import DatasetReader
    stamps = [1700000000000, 1700000000100]
    os.makedirs(tmp_path / "camera_front")
    os.makedirs(tmp_path / "seg_camera_front")
    json.dump({"sync": {"camera_front": stamps}, "unsync": {}, "calibration": {}}, open(tmp_path / "attribute.json", "w"))
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)
    lab = rng.integers(0, 20, (12, 16), dtype=np.uint8)
    for ts in stamps:
        Image.fromarray(rgb).save(tmp_path / "camera_front" / f"{ts}.jpg", quality=100, subsampling=0)
        Image.fromarray(lab).save(tmp_path / "seg_camera_front" / f"{ts}.png")
    dr = DatasetReader(str(tmp_path))
    cams = list(dr.yield_camera("camera_front"))
    assert [t for t, _ in cams] == [s / 1000.0 for s in stamps]
    assert cams[0][1].shape == (12, 16, 3) and np.abs(cams[0][1][..., ::-1].astype(int) - rgb).max() <= 12   # BGR, lossy
    segs = list(dr.yield_semantic("camera_front", start_idx=1))
    assert len(segs) == 1 and segs[0][0] == stamps[1] / 1000.0 and np.array_equal(segs[0][1], lab)
    Image.fromarray(rgb).save(tmp_path / "seg_camera_front" / f"{stamps[0]}.png")
    assert np.array_equal(next(dr.yield_semantic("camera_front"))[1], rgb[..., ::-1])


def test_add_frame_pipes_the_reference_bytes():
    """VideoGenerator.add_frame writes exactly image.astype(uint8).tobytes() (tools.py:28-32), zero-copy when it can."""
    class _Pipe:
        def __init__(self):
            self.data = b""

        def write(self, b):
            self.data += bytes(b)

    class _Writer:
        stdin = _Pipe()

    vg = object.__new__(VideoGenerator)
    vg.writer = _Writer()
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (6, 8, 3), dtype=np.uint8)
    want = b""
    for img in (a, a[:, ::2], a.astype(np.int64)):
        vg.add_frame(img)
        want += img.astype(np.uint8).tobytes()
    assert _Writer.stdin.data == want
    vg.writer = None


def test_video_generator_sink_and_i420_stream(tmp_path):
    """A VideoGenerator writing to a sink: plain arrays first -> a bgr24 stream (the reference's bytes); an object that
    offers I420 planes first -> a yuv420p stream, into which later plain arrays are converted on the host with the same
    arithmetic as the oracle's libswscale restatement."""
    import io
    from oracle import cama_oracle as O
    from cama_amd import runtime
    from cama_amd.egress import bgr_to_i420_host
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (8, 32, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (8, 32, 3), dtype=np.uint8)
    assert np.array_equal(bgr_to_i420_host(a), O.bgr_to_i420(a))
    extremes = np.array([[[0, 0, 0], [255, 255, 255]], [[255, 0, 0], [0, 0, 255]]], np.uint8).repeat(2, 0).repeat(8, 1)
    assert np.array_equal(bgr_to_i420_host(extremes), O.bgr_to_i420(extremes))
    buf = io.BytesIO()
    vg = VideoGenerator(str(tmp_path / "x.mp4"), (32, 8), sink=buf)
    assert runtime.egress_mode() == "bgr24"             # announced itself to the render path; default = the reference's bytes
    vg.add_frame(a)
    assert vg.pix_fmt == "bgr24" and buf.getvalue() == a.tobytes()
    vg.close()

    class Planes:                                        # stands in for egress.DeviceMosaic
        shape, dtype = a.shape, a.dtype

        def i420(self):
            return O.bgr_to_i420(a)

        def __array__(self, dtype=None, copy=None):
            return a

    buf = io.BytesIO()
    runtime.set_egress_format("i420")                   # the opt-in (configs["egress"] = "i420" / CAMA_EGRESS=i420)
    try:
        vg = VideoGenerator(str(tmp_path / "y.mp4"), (32, 8), sink=buf)
        vg.add_frame(Planes())
        vg.add_frame(b)
        assert vg.pix_fmt == "yuv420p"
        assert buf.getvalue() == O.bgr_to_i420(a).tobytes() + O.bgr_to_i420(b).tobytes()
        assert len(buf.getvalue()) == 2 * 8 * 32 * 3 // 2
        assert runtime.egress_mode() == "i420"
        vg.close()
    finally:
        runtime.set_egress_format(None)
    assert runtime.egress_mode() is None                # the render path stops preparing host copies once the sink is closed


def test_frame_poses_are_memoised_per_track_bit_identically(tmp_path, monkeypatch):
    """ClipManager.frame_poses(dataset) is a pure function of the clip's pose track and frame stamps, both parsed once: the
    second call returns the SAME (read-only) arrays, and they equal the recomputation (dataset.POSE_MEMO = False) bit for bit."""
    from cama_amd import dataset as cama_dataset
    from cama_amd.dataset import ClipManager
    g = load_golden("c_gaps")
    clip = rebuild_clip(g, tmp_path)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    for ds in cm.instance_maps:
        a = cm.frame_poses(ds)
        b = cm.frame_poses(ds)
        assert a[0] is b[0] and a[1] is b[1] and not a[1].flags.writeable and a[1].dtype == np.float32
        monkeypatch.setattr(cama_dataset, "POSE_MEMO", False)
        c = cm.frame_poses(ds)
        monkeypatch.setattr(cama_dataset, "POSE_MEMO", True)
        assert c[1] is not a[1] and np.array_equal(c[0], a[0]) and c[1].tobytes() == a[1].tobytes()
        assert a[0].tolist() == g[f"{ds}_frame_ids"].tolist()
        # a re-parsed track (new objects) invalidates the memo
        cm._track_cache.pop(ds)
        d = cm.frame_poses(ds)
        assert d[1] is not a[1] and d[1].tobytes() == a[1].tobytes()
