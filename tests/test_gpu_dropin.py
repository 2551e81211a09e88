"""The drop-in class surface on the GPU: main.py's loop (lazy, fused) and the generic list-of-dicts API,
against the reference's golden vectors and the oracle."""
import numpy as np
import pytest

from oracle import cama_oracle as O
from tests.helpers import (CAMERA_NAMES, DEFAULT_CAMA_CONFIGS, assert_instances_equal, golden_instances, load_golden,
                           rebuild_clip)

pytestmark = pytest.mark.gpu


def _oracle_frames(clip, configs, dataset, static, cams):
    att = O.read_attribute(clip)
    for idx, w2c, cropped in O.iter_frames(clip, att, configs, static, dataset):
        yield idx, w2c, cropped, O.project_all(cropped, cams)


@pytest.mark.parametrize("tag", ["a", "c_gaps", "e_crop"])
def test_lazy_handles_materialise_to_reference_values(tag, tmp_path):
    """yield_frame / project_all_camera results, when touched, equal the reference's lists exactly."""
    from cama.dataset import ClipManager          # the drop-in import path main.py uses
    g = load_golden(tag)
    clip = rebuild_clip(g, tmp_path)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    for ds in cm.instance_maps:
        seen = []
        for image_idx, instance_map in cm.yield_frame(dataset=ds):
            seen.append(image_idx)
            key = f"{ds}_f{image_idx}"
            maps_2d = cm.project_all_camera(instance_map)            # lazy, nothing computed yet
            assert maps_2d._items is None and instance_map._items is None
            for name in CAMERA_NAMES:
                assert_instances_equal(maps_2d[name], golden_instances(g, f"{key}_{name}_vu"))
            assert_instances_equal(list(instance_map), golden_instances(g, key + "_crop"))
            # generic path on the materialised lists gives the same answer
            generic = cm.project_all_camera(list(instance_map))
            for name in CAMERA_NAMES:
                assert_instances_equal(generic[name], golden_instances(g, f"{key}_{name}_vu"))
            # MapManager's list API
            w2c = g[key + "_w2c"]
            t = cm.mm.transform_3d_instance_maps(cm.instance_maps[ds], w2c)
            assert_instances_equal(cm.mm.crop_3d_instance_maps(t), golden_instances(g, key + "_crop"))
        assert seen == g[f"{ds}_frame_ids"].tolist()


def test_main_loop_renders_like_the_oracle(tmp_path):
    """main.py's loop, verbatim, on a clip with real frame files (identity resample): byte-identical mosaics."""
    from cama.dataset import ClipManager
    from cama.tools import VideoGenerator
    from cama_amd.synth import make_clip
    H, W = 96, 160
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=4, seed=5, n_lines=8, verts_per_line=5, line_len_m=3.0, raster_size=400,
              image_mode="npy", image_size=(H, W), origin_size=(H, W))
    configs = dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W))
    cm = ClipManager(configs, clip)
    vg = object.__new__(VideoGenerator)                       # no ffmpeg on the boxes: skip the encoder ctor
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
    for ds in ("cama", "nuscenes"):
        static = cm.instance_maps[ds]
        want = {i: (w2c, m2) for i, w2c, _, m2 in _oracle_frames(clip, configs, ds, static, cams)}
        n = 0
        for image_idx, instance_map in cm.yield_frame(dataset=ds):
            maps_2d_dict = cm.project_all_camera(instance_map)
            image_dict = cm.render_vectors(maps_2d_dict, image_idx)
            image = vg.concate_image(image_dict)
            # oracle: per camera, read the same frame file, stamp with the restated circle, mosaic
            imgs = {}
            for c in cams:
                img = np.load(f"{clip}/{c['name']}/{att['sync'][c['name']][image_idx]}.npy")
                imgs[c["name"]] = O.render_instances(img.copy(), want[image_idx][1][c["name"]])
            ref = O.mosaic(imgs)
            assert image.dtype == np.uint8 and image.shape == ref.shape
            assert np.array_equal(image, ref)
            assert np.array_equal(image_dict["camera_rear"], imgs["camera_rear"])
            assert (ref != O.mosaic({c["name"]: np.load(f"{clip}/{c['name']}/{att['sync'][c['name']][image_idx]}.npy")
                                     for c in cams})).any()          # something was drawn
            # generic render path (plain dicts of arrays) == fused path
            plain = cm.render_vectors({k: list(v) for k, v in maps_2d_dict.items()}, image_idx)
            assert np.array_equal(vg.concate_image(plain), ref)
            n += 1
        assert n == len(want) == 3


def test_reprojector_facade_and_yaml_entry_render_like_the_oracle(tmp_path):
    """The north_star's `Reprojector` (cama.reproject) built from the loaded examples/config.yaml: frames() = main.py:57-60 as
    one generator, project() / render() = the two ClipManager calls -- byte-identical to the oracle on both datasets; and
    examples/run_config.py (the yaml-driven main.py loop) streams exactly those mosaics."""
    import os
    import sys
    from cama.reproject import Reprojector, load_configs
    from cama_amd.synth import make_clip
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    configs = load_configs(os.path.join(repo, "examples", "config.yaml"))
    H, W = 96, 160
    clip = str(tmp_path / "clips" / configs["scene_names"][0])
    make_clip(clip, n_frames=5, seed=11, n_lines=8, verts_per_line=5, line_len_m=3.0, raster_size=400,
              image_mode="npy", image_size=(H, W), origin_size=(H, W))
    rp = Reprojector(configs, clip, output_size=(H, W))
    assert rp.datasets() == ["cama", "nuscenes"]
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
    cc = dict(configs["cama_configs"], output_size=(H, W))
    refs = {}
    for ds in rp.datasets():
        want = {i: m2 for i, _, _, m2 in _oracle_frames(clip, cc, ds, rp.clip.instance_maps[ds], cams)}
        got = list(rp.frames(ds))
        assert [i for i, _ in got] == sorted(want) and len(got) == 4
        for image_idx, mosaic in got:
            imgs = {c["name"]: O.render_instances(
                np.load(f"{clip}/{c['name']}/{att['sync'][c['name']][image_idx]}.npy").copy(), want[image_idx][c["name"]])
                for c in cams}
            ref = O.mosaic(imgs)
            assert isinstance(mosaic, np.ndarray) and np.array_equal(mosaic, ref)
            refs[(ds, image_idx)] = ref
        # the two delegating calls, spelled out
        for image_idx, instance_map in rp.yield_frame(ds):
            image_dict = rp.render(rp.project(instance_map), image_idx)
            assert np.array_equal(rp.mosaic(image_dict), refs[(ds, image_idx)])
            break
    # the yaml-driven loop: same clip (it exists already, so --synthetic leaves it alone), raw bgr24 streams into files
    sys.path.insert(0, os.path.join(repo, "examples"))
    import run_config
    cfg_path = tmp_path / "config.yaml"
    import yaml
    one = dict(configs, scene_names=configs["scene_names"][:1])
    one["cama_configs"] = dict(configs["cama_configs"], output_size=[H, W])
    cfg_path.write_text(yaml.safe_dump(one))
    sink = tmp_path / "streams"
    sink.mkdir()
    done = run_config.main(["-c", str(cfg_path), "--root", str(tmp_path), "--sink", str(sink)])
    assert done[0][0] == configs["scene_names"][0] and done[0][1] == 8
    for ds, suffix in (("cama", "cama"), ("nuscenes", "nuScenes")):
        stream = (sink / f"{configs['scene_names'][0]}_{suffix}.mp4.bgr24").read_bytes()
        assert stream == b"".join(refs[(ds, i)].tobytes() for i in sorted(i for d, i in refs if d == ds))


def test_bgr_to_i420_matches_the_swscale_restatement():
    """cama_bgr_to_i420 (mosaic egress) == the oracle's restatement of libswscale's unscaled BGR24 -> YUV420P C path, byte
    for byte: random frames, the 2880x1080 mosaic size, a strided batch, and the extreme colours."""
    import torch
    from cama_amd import _lib, runtime
    eng = runtime.engine()
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for n, H, W in ((3, 8, 32), (2, 1080, 2880), (1, 90, 160)):
        src = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device=eng.device)
        if (n, H, W) == (3, 8, 32):
            src[0, :, :16] = 255
            src[0, :, 16:] = 0
            src[1, ..., 0], src[1, ..., 1], src[1, ..., 2] = 255, 0, 0
        per = H * W * 3 // 2
        dst = torch.zeros((n, per + 16), dtype=torch.uint8, device=eng.device)       # strided destination
        _lib.check(L.cama_bgr_to_i420(src.data_ptr(), H * W * 3, dst.data_ptr(), per + 16, n, H, W, st))
        torch.cuda.synchronize()
        got, host = dst.cpu().numpy(), src.cpu().numpy()
        for k in range(n):
            assert np.array_equal(got[k, :per], O.bgr_to_i420(host[k])), (n, H, W, k)
            assert not got[k, per:].any()
    assert L.cama_bgr_to_i420(src.data_ptr(), 90 * 160 * 3, dst.data_ptr(), 90 * 160 * 3 // 2, 1, 91, 160, st) == -1   # odd H
    assert L.cama_bgr_to_i420(src.data_ptr(), 90 * 160 * 3, dst.data_ptr(), 90 * 160 * 3 // 2, 1, 90, 150, st) == -1   # W % 16


def test_main_loop_with_video_generator_streams_the_references_bytes_by_default(tmp_path):
    """main.py:56-61 verbatim INCLUDING the VideoGenerator (writing to a sink: no ffmpeg on the boxes).  Frames are
    rendered ahead in batches behind the per-frame surface.

    Default leg: concate_image returns a real ndarray and the pipe carries the reference's bgr24 bytes
    (cama/tools.py:27-32) -- frame by frame the oracle's mosaic -- for every render_ahead setting; the arrays are views of
    the batch's pinned host copy and stay intact while the caller holds them.
    Opt-in leg (configs["egress"] = "i420"): the frames leave the GPU as planar YUV 4:2:0; the stream is the
    libswscale-restated conversion of the oracle's mosaic."""
    import io
    from cama.dataset import ClipManager
    from cama.tools import VideoGenerator
    from cama_amd import runtime
    from cama_amd.egress import DeviceMosaic
    from cama_amd.synth import make_clip
    H, W = 96, 160
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=11, seed=6, n_lines=8, verts_per_line=5, line_len_m=3.0, raster_size=400,
              image_mode="npy", image_size=(H, W), origin_size=(H, W), with_nuscenes=False)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]

    def oracle_mosaics(cm, configs):
        want = {i: m2 for i, _, _, m2 in _oracle_frames(clip, configs, "cama", cm.instance_maps["cama"], cams)}
        out = []
        for image_idx in sorted(want):
            imgs = {}
            for c in cams:
                img = np.load(f"{clip}/{c['name']}/{att['sync'][c['name']][image_idx]}.npy")
                imgs[c["name"]] = O.render_instances(img.copy(), want[image_idx][c["name"]])
            out.append(O.mosaic(imgs))
        return out

    def loop(configs, check_image):
        cm = ClipManager(configs, clip)
        sink = io.BytesIO()
        vg = VideoGenerator(str(tmp_path / "out.mp4"), (3 * W, 2 * H), sink=sink)
        held = []
        for image_idx, instance_map in cm.yield_frame(dataset="cama"):
            maps_2d_dict = cm.project_all_camera(instance_map)
            image_dict = cm.render_vectors(maps_2d_dict, image_idx)
            image = vg.concate_image(image_dict)
            check_image(image)
            vg.add_frame(image)
            held.append(image)
        vg.close()
        return cm, vg, sink.getvalue(), held

    try:
        # ---- default: the reference's bytes
        streams = {}
        for ahead in (4, 1, 16):
            configs = dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W), render_ahead=ahead)

            def is_ndarray(image):
                assert type(image) is np.ndarray and image.shape == (2 * H, 3 * W, 3) and image.dtype == np.uint8
            cm, vg, streams[ahead], held = loop(configs, is_ndarray)
            assert len(held) == 10 and vg.pix_fmt == "bgr24"
        ref = oracle_mosaics(cm, configs)
        per = 2 * H * 3 * W * 3
        assert len(streams[4]) == 10 * per and streams[4] == streams[1] == streams[16]
        for k in range(10):
            assert streams[4][k * per:(k + 1) * per] == ref[k].tobytes(), k
            assert np.array_equal(held[k], ref[k]), k             # every array the loop was handed is still that frame
        # ---- opt-in: device-side I420
        streams = {}
        for ahead in (4, 16):
            configs = dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W), render_ahead=ahead, egress="i420")

            def is_handle(image):
                assert isinstance(image, DeviceMosaic) and image.shape == (2 * H, 3 * W, 3)
            cm, vg, streams[ahead], held = loop(configs, is_handle)
            assert vg.pix_fmt == "yuv420p"
        per = 2 * H * 3 * W * 3 // 2
        assert len(streams[4]) == 10 * per and streams[4] == streams[16]
        for k in range(10):
            assert streams[4][k * per:(k + 1) * per] == O.bgr_to_i420(ref[k]).tobytes(), k
        # touching the lazy mosaic gives the reference's BGR array
        assert np.array_equal(np.asarray(held[-1]), ref[-1]) and held[-1].astype(np.uint8).tobytes() == ref[-1].tobytes()
    finally:
        runtime.set_egress_format(None)
        while runtime.egress_mode() is not None:
            runtime.request_egress(None)


def test_render_clip_batched_equals_per_frame_and_device_source(tmp_path):
    import torch
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import DeviceFrameSource
    from cama_amd.synth import make_clip
    H, W = 64, 112
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=7, seed=9, n_lines=10, verts_per_line=4, line_len_m=2.0, raster_size=400,
              origin_size=(H, W), with_nuscenes=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W)), clip)
    frames = torch.randint(0, 256, (7, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    cm.set_frame_source(DeviceFrameSource(frames))
    idx, mosaic = cm.render_clip("cama")
    idx2, mosaic2 = cm.render_clip("cama", frames_per_launch=2)
    assert idx.tolist() == idx2.tolist() == [1, 2, 3, 4, 5, 6]
    assert torch.equal(mosaic, mosaic2)
    # two-stream pipelined issue (several launches in flight, scratch double-buffered) gives the same bytes
    from cama_amd import runtime
    outs = [torch.zeros_like(mosaic) for _ in range(4)]
    for o in outs:
        cm.render_clip("cama", out=o, frames_per_launch=2, pipelined=True)
    runtime.engine().join()
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, mosaic)
    per_frame = []
    for image_idx, instance_map in cm.yield_frame("cama"):
        per_frame.append(cm.render_vectors(cm.project_all_camera(instance_map), image_idx).mosaic())
    assert np.array_equal(mosaic.cpu().numpy(), np.stack(per_frame))
    # and against the oracle
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
    xyz, col, _, _ = O.flatten_instances(cm.instance_maps["cama"])
    _, w2c = cm.frame_poses("cama")
    src = frames.cpu().numpy()
    for k, i in enumerate(idx):
        flat = O.frame_project_flat(xyz, w2c[k], cams, W, H)
        assert np.array_equal(mosaic[k].cpu().numpy(), O.frame_render_flat(src[i], flat["vu"], flat["vis"], col))


def test_second_pass_over_a_clip_renders_again(tmp_path):
    """The per-frame loop renders `render_ahead` frames per launch and keeps the batches of the current pass; a NEW pass
    (the frame index goes back) must not hand out what the previous pass rendered: the frames may have changed."""
    import torch
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import DeviceFrameSource
    from cama_amd.synth import make_clip
    H, W = 64, 112
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=9, seed=5, n_lines=10, verts_per_line=4, line_len_m=2.0, raster_size=400,
              origin_size=(H, W), with_nuscenes=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W), render_ahead=3), clip)
    frames = torch.randint(0, 256, (9, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    cm.set_frame_source(DeviceFrameSource(frames))

    def one_pass():
        return np.stack([cm.render_vectors(cm.project_all_camera(m), i).mosaic() for i, m in cm.yield_frame("cama")])

    first = one_pass()
    assert np.array_equal(first, one_pass())
    frames.copy_(255 - frames)                                   # same tensor, same source object, new content
    second = one_pass()
    assert not np.array_equal(first, second)
    _, want = cm.render_clip("cama")
    assert np.array_equal(second, want.cpu().numpy())


def _main_loop(cm, dataset):
    """main.py:57-61 with a sink-less VideoGenerator: the mosaics of one pass, copied."""
    from cama.tools import VideoGenerator
    vg = object.__new__(VideoGenerator)
    out = []
    for image_idx, instance_map in cm.yield_frame(dataset=dataset):
        image = vg.concate_image(cm.render_vectors(cm.project_all_camera(instance_map), image_idx))
        out.append(np.array(image, copy=True))
    return out


def test_sweep_reuses_the_process_wide_state_and_the_second_pass_reads_hbm(tmp_path):
    """VERDICT r4 item 2 (the reference's real workload: a fresh ClipManager per scene, both dataset passes once,
    main.py:32-70).  (a) The second pass of a clip takes its frames from HBM (decoded once per clip) -- unless a file changed
    in between, which is seen and decoded again.  (b) The SECOND clip of a sweep lives on what the process-wide Engine
    already owns -- JPEG decoder lanes and pinned arenas, reader threads, pump stream, pooled mosaics: no new device
    segment of mosaic / frame size.  (c) Bytes: both clips, both passes, equal the one-frame-per-launch render."""
    import gc
    import os
    import sys
    import torch
    from PIL import Image
    from cama.dataset import ClipManager
    from cama_amd import runtime
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cold_sweep
    clips = cold_sweep.write_clips(str(tmp_path), 3, 8, distinct=8)
    eng = runtime.engine()

    def sweep_one(clip, check):
        cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
        got = {ds: _main_loop(cm, ds) for ds in ("cama", "nuscenes")}
        stats = dict(cm.frame_source().cache_stats)
        if check:
            ref = ClipManager(dict(DEFAULT_CAMA_CONFIGS, render_ahead=1), clip)      # one frame per launch, no render-ahead
            os.environ["CAMA_FRAME_CACHE_BYTES"] = "0"
            try:
                ref.set_frame_source(None)
                for ds in ("cama", "nuscenes"):
                    want = _main_loop(ref, ds)
                    assert len(want) == len(got[ds]) == 8
                    assert all(np.array_equal(a, b) for a, b in zip(got[ds], want)), ds
                    assert any((a != got[ds][0]).any() for a in got[ds][1:])
            finally:
                del os.environ["CAMA_FRAME_CACHE_BYTES"]
        return cm, stats

    cm, stats = sweep_one(clips[0], check=True)
    # (a) pass 1: 8 frames in batches of 4 + 4 (a short first batch, then up to 16): two misses; pass 2: both from HBM
    assert stats["misses"] == 2 and stats["hits"] == 2 and stats["stored_batches"] == 2 and stats["stale"] == 0, stats
    # a file of frame 2 changes on disk: its batch is decoded again on the next pass, the other one is still served from HBM
    path = cm.cm_list[1].get_image_path(2, True)
    img = np.asarray(Image.open(path)).copy()
    os.unlink(path)                                                # (the clips hard-link their JPEGs: do not write through)
    Image.fromarray(255 - img).save(path, quality=90)
    before = _main_loop(cm, "cama")
    st = cm.frame_source().cache_stats
    assert st["stale"] == 1 and st["misses"] == 3 and st["hits"] == 3, st
    cm2 = ClipManager(dict(DEFAULT_CAMA_CONFIGS, render_ahead=1), clips[0])
    want = _main_loop(cm2, "cama")
    assert all(np.array_equal(a, b) for a, b in zip(before, want))
    del cm, cm2, before, want
    gc.collect()
    # (b) warm process, second and third clip
    sweep_one(clips[1], check=True)
    gc.collect()
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_stats(eng.device)
    pool0 = dict(eng.pool.stats)
    dec = eng.jpeg_decoder()
    lanes0, arenas0 = len(dec._lane), len(dec.__dict__.get("_arenas", []))
    sweep_one(clips[2], check=False)
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_stats(eng.device)
    grown = m1["reserved_bytes.all.allocated"] - m0["reserved_bytes.all.allocated"]
    assert grown < (32 << 20), f"the third clip of a sweep made torch reserve {grown / 1e6:.0f} MB of new device segments"
    assert eng.pool.stats["allocations"] == pool0["allocations"] and eng.pool.stats["hits"] > pool0["hits"]
    assert eng.jpeg_decoder() is dec and len(dec._lane) == lanes0 and len(dec.__dict__.get("_arenas", [])) == arenas0


def test_crop_dict_override_is_honoured(tmp_path):
    from cama_amd.dataset import ClipManager
    g = load_golden("e_crop")
    clip = rebuild_clip(g, tmp_path)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    cm.mm.crop_dict["x_max"] = 10
    cm.mm.crop_dict["x_min"] = -10
    for image_idx, instance_map in cm.yield_frame("nuscenes"):
        pts = np.concatenate([i["points"] for i in instance_map])
        assert pts[:, 0].max() <= 10 and pts[:, 0].min() >= -10 and len(pts) > 0
        full = golden_instances(g, f"nuscenes_f{image_idx}_crop")
        assert len(pts) < sum(p.shape[0] for _, p in full)
        break


def test_default_output_size_flow_with_jpeg_frames(tmp_path):
    """The reference's default CameraManager output (540,960) from 900x1600 JPEGs: decode (Pillow) -> device
    resample -> fused overlay, vs the oracle's restated remap + circle on the same decoded frames."""
    from cama.dataset import ClipManager
    from cama.tools import VideoGenerator
    from cama_amd.frames import read_bgr
    from cama_amd.synth import make_clip
    clip = str(tmp_path / "clip")
    make_clip(clip, n_frames=2, seed=6, n_lines=6, verts_per_line=5, line_len_m=3.0, raster_size=400,
              image_mode="jpg", image_size=(900, 1600), with_nuscenes=False)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)                  # default output_size (540, 960)
    assert (cm.cm_list[0].height, cm.cm_list[0].width) == (540, 960)
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n) for n in CAMERA_NAMES]
    vg = object.__new__(VideoGenerator)
    (image_idx, instance_map), = list(cm.yield_frame("cama"))
    image = vg.concate_image(cm.render_vectors(cm.project_all_camera(instance_map), image_idx))
    assert image.shape == (1080, 2880, 3)
    want = {i: m2 for i, _, _, m2 in _oracle_frames(clip, cm.configs, "cama", cm.instance_maps["cama"], cams)}
    imgs = {}
    for c in cams:
        raw = read_bgr(f"{clip}/{c['name']}/{att['sync'][c['name']][image_idx]}.jpg")
        assert raw.shape == (900, 1600, 3)
        # separable zero-distortion map: build one row / one column with the oracle and broadcast
        mx, _ = O.undistort_map(c["K_origin"], c["d_origin"], c["K"], 960, 1)
        _, my = O.undistort_map(c["K_origin"], c["d_origin"], c["K"], 1, 540)
        small = O.remap_bilinear(raw, np.broadcast_to(mx, (540, 960)), np.broadcast_to(my, (540, 960)))
        imgs[c["name"]] = O.render_instances(small, want[image_idx][c["name"]])
    assert np.array_equal(image, O.mosaic(imgs))
    # the per-camera host API gives the same resized frame
    cam0 = cm.cm_list[1]
    assert np.array_equal(cam0.read_resized_image_by_index(image_idx),
                          O.remap_bilinear(read_bgr(cam0.get_image_path(image_idx, True)),
                                           *[np.broadcast_to(m, (540, 960)) for m in
                                             (O.undistort_map(cam0.K_origin, cam0.d_origin, cam0.K, 960, 1)[0],
                                              O.undistort_map(cam0.K_origin, cam0.d_origin, cam0.K, 1, 540)[1])]))


@pytest.mark.parametrize("tag", ["a", "d_nusonly"])
def test_device_static_map_build_is_bit_identical(tag, tmp_path):
    """configs["device_map_build"]: densify + height gather + pixel->world on the GPU == the reference's static map."""
    from cama_amd.dataset import ClipManager, StaticInstances
    g = load_golden(tag)
    clip = rebuild_clip(g, tmp_path)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, device_map_build=True), clip)
    host = ClipManager(dict(DEFAULT_CAMA_CONFIGS), clip)
    for ds in cm.instance_maps:
        ins = cm.instance_maps[ds]
        assert isinstance(ins, StaticInstances) and ins._items is None
        # fused path straight from the device-built buffer (no host copy yet)
        for (i, fm), (j, hm) in zip(cm.yield_frame(ds), host.yield_frame(ds)):
            assert i == j
            a, b = cm.project_all_camera(fm), host.project_all_camera(hm)
            for name in CAMERA_NAMES:
                assert_instances_equal(a[name], golden_instances(g, f"{ds}_f{i}_{name}_vu"))
                assert_instances_equal(b[name], golden_instances(g, f"{ds}_f{i}_{name}_vu"))
        # ... and the public list, when touched, is the reference's static map, bit for bit
        assert_instances_equal(list(ins), golden_instances(g, f"{ds}_static"))
        assert all(p["points"].dtype == np.float32 for p in ins)


def test_device_static_map_float64_raster_and_edge_pixels(tmp_path):
    import json
    from cama_amd import runtime
    from cama_amd.reproject import MapManager
    rng = np.random.default_rng(2)
    mm = MapManager()
    bev = rng.normal(0, 0.1, (50, 50))                        # float64 raster -> float64 points
    labels = [{"attrs": {"type": "lane_marking"}, "data": [[-3.2, 5.0], [4.0, 9.0], [4.02, 9.01], [60.0, 70.0]]},
              {"attrs": {"type": "Road_teeth"}, "data": [[48.4, 0.5], [49.5, 2.5], [51.5, 48.5]]},
              {"attrs": {"type": "x"}, "data": [[1, 1]]}]
    want = mm.calculate_3d_instance_maps(bev, labels)
    table = mm.segment_table(labels)
    dmap = runtime.engine().build_static_map(table, lift=True, bev_height=bev)
    assert dmap.is_f64 == 1 and dmap.N == sum(w["points"].shape[0] for w in want)
    got = dmap.soa.cpu().numpy().T
    assert np.array_equal(got, np.concatenate([w["points"] for w in want]))
    want32 = mm.calculate_3d_instance_maps(bev.astype(np.float32), labels)
    d32 = runtime.engine().build_static_map(table, lift=True, bev_height=bev.astype(np.float32))
    assert np.array_equal(d32.soa.cpu().numpy().T, np.concatenate([w["points"] for w in want32]))
    assert d32.colour.cpu().numpy().tolist() == [0] * want[0]["points"].shape[0] + [1] * want[1]["points"].shape[0]
    # rasters that are neither float32 nor float64 (float16 / integer .npy): the vertex dtype follows numpy's promotion
    # of (float32, raster dtype) like the reference's np.concatenate, decided BEFORE the vertex buffer is sized (a
    # float32-sized buffer written with doubles overran by 4*N bytes)
    import torch
    for odd, is64 in ((bev.astype(np.float16), 0), ((bev * 100).astype(np.int16), 0), ((bev * 1e6).astype(np.int32), 1)):
        want_odd = np.concatenate([w["points"] for w in mm.calculate_3d_instance_maps(odd, labels)])
        assert want_odd.dtype == (np.float64 if is64 else np.float32)
        d_odd = runtime.engine().build_static_map(table, lift=True, bev_height=odd)
        assert d_odd.is_f64 == is64 and d_odd.soa.dtype == (torch.float64 if is64 else torch.float32)
        assert np.array_equal(d_odd.soa.cpu().numpy().T, want_odd)
        _ = torch.zeros(8, device="cuda").sum().item()          # the device is still healthy (no overrun)


def test_fused_raw_frame_overlay_equals_resample_then_overlay(tmp_path):
    """cama_overlay_frames_raw (undistort+resize fused into the overlay's source read) == cama_resample_frames followed
    by the plain overlay, byte for byte; with and without lens distortion (separable / full 2-D maps)."""
    import torch
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import RawDeviceFrameSource
    from cama_amd.synth import make_clip
    for k, dist in enumerate((False, True)):
        clip = str(tmp_path / f"clip{k}")
        make_clip(clip, n_frames=4, seed=20 + k, n_lines=10, verts_per_line=5, line_len_m=3.0, raster_size=400,
                  origin_size=(90, 160), with_nuscenes=False, d_nonzero=dist)
        H, W = 54, 96                                                   # 0.6 scale like 900x1600 -> 540x960
        outs = []
        raw = torch.randint(0, 256, (4, 6, 90, 160, 3), dtype=torch.uint8, device="cuda")
        for fused in (True, False):
            cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W)), clip)
            assert cm.cm_list[0].needs_resample()
            cm.set_frame_source(RawDeviceFrameSource(raw, cm.cm_list, fused=fused))
            idx, mosaic = cm.render_clip("cama")
            torch.cuda.synchronize()
            outs.append(mosaic.clone())
        assert idx.tolist() == [1, 2, 3] and tuple(outs[0].shape) == (3, 2 * H, 3 * W, 3)
        assert torch.equal(outs[0], outs[1])
        # against the oracle's restated remap + circle
        att = O.read_attribute(clip)
        cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
        xyz, col, _, _ = O.flatten_instances(cm.instance_maps["cama"])
        _, w2c = cm.frame_poses("cama")
        rawh = raw.cpu().numpy()
        small = np.stack([O.remap_bilinear(rawh[2, c], *O.undistort_map(cam["K_origin"], cam["d_origin"], cam["K"], W, H))
                          for c, cam in enumerate(cams)])
        flat = O.frame_project_flat(xyz, w2c[1], cams, W, H)
        assert flat["vis"].sum() > 100
        assert np.array_equal(outs[0][1].cpu().numpy(), O.frame_render_flat(small, flat["vu"], flat["vis"], col))


def _raw_pipeline_vs_oracle(tmp_path, origin, out_size, seed, n_frames=4, check_frames=(1,), n_lines=12, min_stamped=100):
    """Fused raw overlay of a zero-distortion clip == resample->overlay == the oracle's restated remap + circles.
    Returns which kernel family the engine selected: "raw35" (3:5 rational pattern), "lds" or "gather"."""
    import torch
    from cama_amd import runtime
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import RawDeviceFrameSource
    from cama_amd.synth import frame_pattern, make_clip
    clip = str(tmp_path / f"clip_{origin[1]}_{out_size[1]}_{n_lines}")
    make_clip(clip, n_frames=n_frames, seed=seed, n_lines=n_lines, verts_per_line=5, line_len_m=3.0, raster_size=400,
              origin_size=origin, with_nuscenes=False)
    H, W = out_size
    raw = frame_pattern(seed, (n_frames, 6, origin[0], origin[1], 3), "cuda")
    outs = []
    for fused in (True, False):
        cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(H, W)), clip)
        cm.set_frame_source(RawDeviceFrameSource(raw, cm.cm_list, fused=fused))
        idx, mosaic = cm.render_clip("cama")
        torch.cuda.synchronize()
        outs.append(mosaic.clone())
        if fused:
            maps = runtime.engine().rig_maps(cm.cm_list)
            kind = "raw35" if maps[-1] is not None else "lds" if (maps[2] and maps[3] is not None) else "gather"
    assert torch.equal(outs[0], outs[1])
    att = O.read_attribute(clip)
    cams = [O.camera_model(att, n, output_size=(H, W)) for n in CAMERA_NAMES]
    xyz, col, _, _ = O.flatten_instances(cm.instance_maps["cama"])
    _, w2c = cm.frame_poses("cama")
    rawh = raw.cpu().numpy()
    stamped = 0
    for k in check_frames:
        small = np.stack([O.remap_bilinear(rawh[idx[k], c], *O.undistort_map(cam["K_origin"], cam["d_origin"], cam["K"], W, H))
                          for c, cam in enumerate(cams)])
        flat = O.frame_project_flat(xyz, w2c[k], cams, W, H)
        stamped += int(flat["vis"].sum())
        assert np.array_equal(outs[0][k].cpu().numpy(), O.frame_render_flat(small, flat["vu"], flat["vis"], col)), k
    assert stamped > min_stamped
    return kind


def test_raw_overlay_kernel_families(tmp_path):
    """The three raw-frame overlays on zero-distortion clips: 160x90 -> 96x54 and 320x180 -> 192x108 are the reference's
    3:5 scale (gather-free k_overlay_raw35, several units per row, bands with and without stamps); 160x90 -> 80x45
    (1:2) is separable but not 3:5 (LDS-staged k_overlay_rawlds); every result equals the oracle."""
    assert _raw_pipeline_vs_oracle(tmp_path, (90, 160), (54, 96), seed=31, check_frames=(0, 1, 2)) == "raw35"
    assert _raw_pipeline_vs_oracle(tmp_path, (180, 320), (108, 192), seed=32) == "raw35"
    assert _raw_pipeline_vs_oracle(tmp_path, (90, 160), (45, 80), seed=33) == "lds"
    # a dense map at 3:5: nearly every band is stamped, many with more stamps than the workgroup has threads (the owner
    # table of a stamped band lives inside k_overlay_raw35's staging area: built after the taps are read)
    assert _raw_pipeline_vs_oracle(tmp_path, (180, 320), (108, 192), seed=35, n_lines=600, check_frames=(0, 2),
                                   min_stamped=20000) == "raw35"


def test_raw_overlay_reference_default_size(tmp_path):
    """The reference's default pipeline at full size: raw 1600x900 frames -> 960x540 tiles -> 2880x1080 mosaic."""
    assert _raw_pipeline_vs_oracle(tmp_path, (900, 1600), (540, 960), seed=34, n_frames=3, check_frames=(1,)) == "raw35"


def test_integration_md_binding_example_runs(repo_root):
    """The ctypes stub shown in INTEGRATION.md (what a maintainer of the reference would add) is real code: extract
    it, point it at the built library, render with it and compare with the engine."""
    import re
    import torch
    from cama_amd import _lib
    from cama_amd.engine import Engine
    from tests.test_gpu_kernels import _random_scene, _rig
    text = open(f"{repo_root}/INTEGRATION.md").read()
    code = re.search(r"```python\n(# cama/_hip\.py.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libcama_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    W, H, N, F = 160, 96, 3000, 2
    xyz, col, cams, w2c = _random_scene(91, N, F, W, H)
    e = Engine("cuda:0")
    rig = _rig(e, cams)
    src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    want = e.render_frames(e.upload_map(xyz, col, spatial_sort=False), rig, w2c, src)
    out = torch.empty_like(want)
    ns["render_frames"](torch.from_numpy(np.ascontiguousarray(xyz.T)).cuda(), torch.from_numpy(col).cuda(),
                        torch.from_numpy(np.asarray(w2c, np.float64).reshape(F, 16)).cuda(), rig.c2cam, rig.K,
                        [-50, 50, -100, 100, -200, 200], src, out)
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_pipelined_raw35_two_rigs_back_to_back(tmp_path):
    """Two clips with DIFFERENT rigs (sizes 160x90 -> 96x54 and 320x180 -> 192x108) rendered alternately through the
    pipelined 3:5 raw overlay with no join in between: the overlay of launch k runs while the host has long since built
    the other rig's tap tables.  A launch keeps its own tables alive; the plan cache is keyed on calibration content,
    not on object ids."""
    import torch
    from cama_amd import runtime
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import RawDeviceFrameSource
    from cama_amd.synth import frame_pattern, make_clip
    eng = runtime.engine()
    clips = []
    for k, (origin, out_size) in enumerate((((90, 160), (54, 96)), ((180, 320), (108, 192)))):
        clip = str(tmp_path / f"clip{k}")
        make_clip(clip, n_frames=5, seed=40 + k, n_lines=40, verts_per_line=5, line_len_m=3.0, raster_size=400,
                  origin_size=origin, with_nuscenes=False)
        raw = frame_pattern(40 + k, (5, 6, origin[0], origin[1], 3), "cuda")
        cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=out_size), clip)
        cm.set_frame_source(RawDeviceFrameSource(raw, cm.cm_list, fused=True))
        _, want = cm.render_clip("cama")                         # plain, single stream
        torch.cuda.synchronize()
        clips.append((cm, raw, want.clone(), torch.zeros_like(want)))
    plans = eng.__dict__["_rig_plans"]
    assert len(plans) >= 2
    for rounds in range(6):
        for cm, raw, want, out in clips:
            cm.render_clip("cama", out=out, pipelined=True)
            # a fresh ClipManager of the same clip (new objects, same calibration) must hit the same plan
            cm2 = ClipManager(dict(cm.configs), cm.clip_path)
            n = len(plans)
            eng.rig_maps(cm2.cm_list)
            assert len(plans) == n
            del cm2
    eng.join()
    torch.cuda.synchronize()
    for cm, raw, want, out in clips:
        assert torch.equal(out, want)
    # evicting every plan while launches are queued must not matter either: the launches hold their tables
    for cm, raw, want, out in clips:
        out.zero_()
        cm.render_clip("cama", out=out, pipelined=True)
        plans.clear()
        junk = [torch.full((1 << 16,), 7, dtype=torch.int32, device="cuda") for _ in range(8)]   # reuse freed blocks
    eng.join()
    torch.cuda.synchronize()
    del junk
    for cm, raw, want, out in clips:
        assert torch.equal(out, want)


def test_main_loop_over_a_clip_with_pose_gaps_and_raw_frames(tmp_path):
    """A clip whose pose track has a gap (frames skipped like cama/dataset.py:93-96) through main.py's loop with raw
    device frames and render-ahead batches: batches end at the hole, every yielded frame equals its one-frame render."""
    import torch
    from cama_amd.dataset import ClipManager
    from cama_amd.frames import RawDeviceFrameSource
    from cama_amd.synth import frame_pattern
    g = load_golden("c_gaps")
    clip = rebuild_clip(g, tmp_path)
    cm = ClipManager(dict(DEFAULT_CAMA_CONFIGS, output_size=(540, 960), render_ahead=4), clip)
    c0 = cm.cm_list[0]
    assert (c0.height_origin, c0.width_origin) == (900, 1600)     # the golden clip is calibrated for 1600x900 frames
    n_img = len(cm._track("cama")[1])
    raw = frame_pattern(3, (n_img, 6, 900, 1600, 3), "cuda")
    cm.set_frame_source(RawDeviceFrameSource(raw, cm.cm_list, fused=True))
    idx, w2c = cm.frame_poses("cama")
    assert len(idx) < n_img - 1 and (np.diff(idx) > 1).any()      # there IS a hole
    seen = []
    for image_idx, instance_map in cm.yield_frame(dataset="cama"):
        image_dict = cm.render_vectors(cm.project_all_camera(instance_map), image_idx)
        got = image_dict.mosaic_device.clone()
        k = int(np.flatnonzero(idx == image_idx)[0])
        _, one = cm.render_clip("cama", poses=(idx[k:k + 1], w2c[k:k + 1]))
        torch.cuda.synchronize()
        assert torch.equal(got, one[0]), image_idx
        seen.append(image_idx)
    assert seen == idx.tolist()
    _, whole = cm.render_clip("cama")                              # the whole-clip path gathers across the hole too
    torch.cuda.synchronize()
    assert whole.shape[0] == len(idx)
