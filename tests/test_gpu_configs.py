"""-m gpu parity for the multi-scene sweep (BASELINE configs[2]) and the 1e6-vertex stress (configs[4]).

The expected values are oracle renders: tests/golden/scene_hashes.json holds shard.overlay_hash of the ORACLE's mosaics
for bench.py's seeded scenes (tests/golden/gen_scene_hashes.py, CPU only), and a few scenes / frames are additionally
compared byte for byte with the oracle here.  The pipelined multi-scene path is exercised the way bench.py --scenes 73
drives it: many distinct scenes back to back, no join in between, the host far ahead of the GPU."""
import argparse
import os

import numpy as np
import pytest

import bench
from cama_amd import shard
from oracle import cama_oracle as O
from tests.golden import gen_scene_hashes as G

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_hashes.json")


def _args(**kw):
    a = bench.parse_args([])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _golden(a, unit="scene"):
    key = bench.workload_key(a.frames, a.verts, a.width, a.height, a.map, unit=unit)
    g = shard.load_golden_hashes(GOLDEN, key)
    assert g, f"no golden entry for {key}"
    return g


def _render_pipelined_no_join(scenes, out, rounds):
    """Every scene into its own slice of `out`, `rounds` times over, pipelined, never joined until the end."""
    import torch
    from cama_amd import runtime
    eng = runtime.engine()
    for _ in range(rounds):
        for k, (cm, _, _) in enumerate(scenes):
            cm.render_clip("cama", out=out[k], pipelined=True)
    issued = int(eng.lib.cama_pipeline_issued(eng._pipe["handle"]))
    eng.join()
    torch.cuda.synchronize()
    return issued


def test_small_sweep_24_distinct_scenes_pipelined_equals_oracle():
    """24 distinct scenes (own map, calibration, poses, frames), pipelined back to back with no join: every scene's
    hash equals the oracle's (golden) and its unpipelined render; three scenes byte-equal to the oracle."""
    import torch
    from cama_amd import runtime
    a = _args(frames=6, verts=3000, height=180, width=320)
    dev = torch.device("cuda:0")
    golden = _golden(a)
    scenes = [bench.build_scene(a, s, dev) for s in range(24)]
    eng = runtime.engine()
    shape = eng.mosaic_shape(scenes[0][0]._rig(), a.frames)
    out = torch.zeros((24,) + tuple(shape), dtype=torch.uint8, device=dev)
    _render_pipelined_no_join(scenes, out, rounds=4)          # 96 launches > the pipeline's 64-event ring
    for k, (cm, _, _) in enumerate(scenes):
        assert shard.overlay_hash(out[k]) == golden[k], f"scene {k}: pipelined render differs from the oracle"
        _, plain = cm.render_clip("cama")
        torch.cuda.synchronize()
        assert torch.equal(plain, out[k]), f"scene {k}: pipelined != plain"
    for k in (0, 11, 23):
        xyz, col, cams, w2c = G.scene_setup(a, k)
        got = out[k].cpu().numpy()
        for pos in range(a.frames):
            assert np.array_equal(got[pos], G.render_frame(a, k, pos, xyz, col, cams, w2c)), (k, pos)


@pytest.fixture(scope="module")
def sweep_scenes():
    import torch
    a = _args()                                               # BASELINE configs[1]/[2] scene: 40 frames, 1600x900
    dev = torch.device("cuda:0")
    return a, [bench.build_scene(a, s, dev) for s in range(24)]


def test_fullsize_sweep_24_scenes_pipelined_run_ahead(sweep_scenes):
    """configs[2] at full size, first 24 scenes: the overlay of a launch lasts ~340 us while the host issues one in
    ~150 us, so the host runs dozens of launches ahead; what a launch reads (its poses!) must survive until that launch
    has run.  Every scene's hash must equal the oracle's render of that scene (golden), after three un-joined rounds."""
    import torch
    from cama_amd import runtime
    a, scenes = sweep_scenes
    golden = _golden(a)
    eng = runtime.engine()
    dev = torch.device("cuda:0")
    shape = eng.mosaic_shape(scenes[0][0]._rig(), a.frames)
    out = torch.zeros((len(scenes),) + tuple(shape), dtype=torch.uint8, device=dev)
    issued0 = int(eng.lib.cama_pipeline_issued(eng._pipeline()["handle"]))
    issued = _render_pipelined_no_join(scenes, out, rounds=3)
    assert issued - issued0 == 3 * len(scenes)
    assert int(eng.lib.cama_pipeline_completed(eng._pipe["handle"])) == issued
    assert not eng._pipe["keep"]
    bad = [k for k in range(len(scenes)) if shard.overlay_hash(out[k]) != golden[k]]
    assert not bad, f"scenes {bad}: pipelined full-size render differs from the oracle's"
    # allocator churn between launches must not matter either: interleave fresh allocations with un-joined launches
    out.zero_()
    junk = []
    for k, (cm, _, _) in enumerate(scenes):
        cm.render_clip("cama", out=out[k], pipelined=True)
        junk.append(torch.full((4096,), float(k), dtype=torch.float64, device=dev))      # same size class as the poses
        junk = junk[-2:]
    eng.join()
    torch.cuda.synchronize()
    bad = [k for k in range(len(scenes)) if shard.overlay_hash(out[k]) != golden[k]]
    assert not bad, f"scenes {bad} after allocator churn"


def test_fullsize_scene_byte_equal_to_oracle_frames(sweep_scenes):
    """Scene 17 of the sweep (never rendered by any other test), sampled frames byte for byte against the oracle."""
    import torch
    a, scenes = sweep_scenes
    cm = scenes[17][0]
    _, mosaic = cm.render_clip("cama")
    torch.cuda.synchronize()
    xyz, col, cams, w2c = G.scene_setup(a, 17)
    for pos in (0, 13, 39):
        assert np.array_equal(mosaic[pos].cpu().numpy(), G.render_frame(a, 17, pos, xyz, col, cams, w2c)), pos


def test_stress_1e6_random_vertices_125_frames():
    """configs[4], one rank's share (125 frames) at a reduced image size: 1e6 random vertices (Morton-sorted copy +
    draw keys + block cull inside).  All frames: every changed pixel lies in a visible point's disc and carries a palette
    colour, every visible point's centre pixel is stamped, renders are deterministic and pipelined == plain; sampled
    frames equal the oracle (golden hashes + two frames byte for byte)."""
    import torch
    from cama_amd import runtime
    a = _args(frames=125, verts=1000000, height=180, width=320, map="random")
    dev = torch.device("cuda:0")
    golden = _golden(a, unit="frame")
    cm, frames, _ = bench.build_scene(a, 0, dev)
    dmap = cm._static("cama").device()
    assert dmap.N == 1000000 and dmap.sorted_soa is not None
    idx, mosaic = cm.render_clip("cama")
    torch.cuda.synchronize()
    assert len(idx) == 125
    for pos, want in golden.items():
        assert shard.overlay_hash(mosaic[pos]) == want, f"frame {pos} differs from the oracle"
    xyz, col, cams, w2c = G.scene_setup(a, 0)
    for pos in (1, 93):
        assert np.array_equal(mosaic[pos].cpu().numpy(), G.render_frame(a, 0, pos, xyz, col, cams, w2c)), pos
    H, W = a.height, a.width
    got_all = mosaic.cpu().numpy()
    src_all = frames.cpu().numpy()
    grey, gold = np.array([211, 211, 211], np.uint8), np.array([0, 215, 255], np.uint8)
    stamped = 0
    for pos in range(125):
        flat = O.frame_project_flat(xyz, w2c[pos], cams, W, H)
        for c in range(6):
            r, q = divmod(c, 3)
            cell = got_all[pos, r * H:(r + 1) * H, q * W:(q + 1) * W]
            changed = (cell != src_all[idx[pos], c]).any(axis=-1)
            vis = flat["vis"][c].astype(bool)
            p = flat["vu"][c][vis].astype(np.int32)
            allowed = np.zeros((H, W), bool)
            for dy in range(-2, 3):
                hw = [2, 1, 0][abs(dy)]
                for dx in range(-hw, hw + 1):
                    y, x = p[:, 0] + dy, p[:, 1] + dx
                    ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
                    allowed[y[ok], x[ok]] = True
            assert not (changed & ~allowed).any(), (pos, c)
            px = cell[allowed]                                  # every footprint pixel carries a palette colour
            assert ((px == grey).all(axis=-1) | (px == gold).all(axis=-1)).all(), (pos, c)
            stamped += len(p)
    assert stamped > 125 * 1000
    h0 = shard.overlay_hash(mosaic)
    _, again = cm.render_clip("cama")
    torch.cuda.synchronize()
    assert shard.overlay_hash(again) == h0
    out = torch.zeros_like(mosaic)
    eng = runtime.engine()
    for lo in range(0, 125, 25):                                # five un-joined pipelined launches of 25 frames
        cm.render_clip("cama", out=out[lo:lo + 25], pipelined=True, poses=(idx[lo:lo + 25], cm.frame_poses("cama")[1][lo:lo + 25]))
    eng.join()
    torch.cuda.synchronize()
    assert torch.equal(out, mosaic)


def test_fullsize_sweep_remaining_scenes_24_to_72():
    """configs[2], the other 49 scenes of the 73-scene sweep at full size (1600x900, 40 frames), one after the other
    (build -> render pipelined -> hash -> free): with the 24 scenes above the driver-run suite covers all 73 golden
    (oracle-rendered) scene hashes, not only the builder's own bench run."""
    import torch
    from cama_amd import runtime
    a = _args()
    golden = _golden(a)
    assert len(golden) == bench.SWEEP_SCENES
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    out = None
    bad = []
    for sid in range(24, bench.SWEEP_SCENES):
        cm, frames, _ = bench.build_scene(a, sid, dev)
        if out is None:
            out = torch.empty(eng.mosaic_shape(cm._rig(), a.frames), dtype=torch.uint8, device=dev)
        out.zero_()
        cm.render_clip("cama", out=out, pipelined=True)
        eng.join()
        torch.cuda.synchronize()
        if shard.overlay_hash(out) != golden[sid]:
            bad.append(sid)
        del cm, frames
    assert not bad, f"scenes {bad}: full-size render differs from the oracle's"


def test_fullsize_stress_sampled_frames_equal_the_oracle():
    """configs[4] at FULL size: 1e6 random vertices x 1000 frames at 1600x900, the 16 sampled frame positions (first and
    last frame of every rank's range for 1, 2, 4 and 8 ranks) against the oracle-rendered golden hashes -- what
    bench.py --gpus N checks in its nested stress, here inside the driver-run suite.  All 1001 frames are resident
    (26 GB of the 288 GB); only the sampled positions are rendered, one launch each and once more as 8-frame launches
    that contain them (a frame must not depend on its launch-mates)."""
    import torch
    from cama_amd import runtime
    a = _args(frames=bench.STRESS["frames"], verts=bench.STRESS["verts"], map="random")
    golden = _golden(a, unit="frame")
    samples = bench.stress_sample_frames(a.frames)
    assert sorted(golden) == samples and len(samples) == 16
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    cm, frames, _ = bench.build_scene(a, 0, dev)
    idx, w2c = cm.frame_poses("cama")
    assert len(idx) == a.frames
    one = torch.empty(eng.mosaic_shape(cm._rig(), 1), dtype=torch.uint8, device=dev)
    for pos in samples:
        one.zero_()
        cm.render_clip("cama", out=one, poses=(idx[pos:pos + 1], w2c[pos:pos + 1]))
        torch.cuda.synchronize()
        assert shard.overlay_hash(one[0]) == golden[pos], f"stress frame {pos} differs from the oracle's"
    many = torch.empty(eng.mosaic_shape(cm._rig(), 8), dtype=torch.uint8, device=dev)
    for pos in samples:
        lo = min(max(0, pos - 3), a.frames - 8)
        many.zero_()
        cm.render_clip("cama", out=many, poses=(idx[lo:lo + 8], w2c[lo:lo + 8]), pipelined=True)
        eng.join()
        torch.cuda.synchronize()
        assert shard.overlay_hash(many[pos - lo]) == golden[pos], f"stress frame {pos} (in an 8-frame launch)"


def test_site_aggregated_scenes_share_one_device_map():
    """configs[3] as SURVEY.md D6 defines it: several scenes driven on ONE site map (2 sites x 4 scenes, 60 000 vertices
    per site, every scene its own pose track and calibration).  Each ClipManager carries its own host copy of the site's
    labels (main.py:42 reads them per scene); on the device they are ONE vertex buffer per site -- uploaded, sorted and
    indexed once -- and every scene's mosaic hash equals the oracle's render of that scene."""
    import torch
    from cama_amd import runtime
    a = _args(frames=6, verts=60000, height=180, width=320, map="site", sites=2, scenes=8)
    key = bench.args_key(a)
    golden = shard.load_golden_hashes(GOLDEN, key)
    assert len(golden) == 8, key
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    before = dict(getattr(eng, "map_cache_stats", {"uploads": 0, "hits": 0}))
    scenes = [bench.build_scene(a, s, dev) for s in range(8)]
    out = torch.zeros((8,) + tuple(eng.mosaic_shape(scenes[0][0]._rig(), a.frames)), dtype=torch.uint8, device=dev)
    for k, (cm, _, _) in enumerate(scenes):
        cm.render_clip("cama", out=out[k], pipelined=True)
    eng.join()
    torch.cuda.synchronize()
    stats = eng.map_cache_stats
    assert stats["uploads"] - before["uploads"] == 2, stats          # one per site, not one per scene
    assert stats["hits"] - before["hits"] == 6, stats
    dmaps = [cm._static("cama").device() for cm, _, _ in scenes]
    for k in range(8):
        assert dmaps[k] is dmaps[k % 2] and dmaps[k].N == 60000
        assert scenes[k][0].instance_maps["cama"] is not scenes[k % 2][0].instance_maps["cama"] or k < 2
    assert dmaps[0] is not dmaps[1] and dmaps[0].soa.data_ptr() != dmaps[1].soa.data_ptr()
    bad = [k for k in range(8) if shard.overlay_hash(out[k]) != golden[k]]
    assert not bad, f"scenes {bad}: render on the shared site map differs from the oracle's"
    # the tracks really differ: scenes of one site do not render the same pixels
    assert len({shard.overlay_hash(out[k]) for k in range(8)}) == 8
    # the cache is weak: dropping every scene of site 1 frees its device map
    import gc
    import weakref
    ref = weakref.ref(dmaps[1])
    del dmaps, cm                                   # (`cm`: the loop variable above still names scene 7, a site-1 scene)
    scenes = [sc for k, sc in enumerate(scenes) if k % 2 == 0]
    eng.join()
    eng.__dict__.pop("_last_bin", None)
    gc.collect()
    assert ref() is None


def test_many_scenes_in_one_launch_equal_the_per_scene_renders():
    """dataset.render_clips: 24 distinct small scenes (own maps with different vertex counts, own calibrations, poses and
    frames) as ONE multi-scene launch chain (cama_pipeline_render_scenes / cama_render_scenes) -- every scene's hash equals
    the oracle's golden hash and the bytes of its own single-scene render; pipelined over several un-joined rounds, on the
    current stream, and cut into groups of a few scenes per launch."""
    import torch
    from cama_amd import runtime
    from cama_amd.dataset import render_clips
    a = _args(frames=6, verts=3000, height=180, width=320)
    dev = torch.device("cuda:0")
    golden = _golden(a)
    scenes = [bench.build_scene(a, s, dev) for s in range(24)]
    cms = [cm for cm, _, _ in scenes]
    assert len({cm._static("cama").device().N for cm in cms}) > 1          # the scenes' maps differ in size
    eng = runtime.engine()
    shape = eng.mosaic_shape(cms[0]._rig(), a.frames)
    outs = [torch.zeros(shape, dtype=torch.uint8, device=dev) for _ in cms]
    issued0 = int(eng.lib.cama_pipeline_issued(eng._pipeline()["handle"]))
    for _ in range(5):
        assert render_clips(cms, "cama", outs, pipelined=True) is True
    assert int(eng.lib.cama_pipeline_issued(eng._pipe["handle"])) - issued0 == 5   # one launch chain per round, not 24
    eng.join()
    torch.cuda.synchronize()
    for k, cm in enumerate(cms):
        assert shard.overlay_hash(outs[k]) == golden[k], f"scene {k}: multi-scene launch differs from the oracle"
        _, plain = cm.render_clip("cama")
        torch.cuda.synchronize()
        assert torch.equal(plain, outs[k]), k
    # current-stream variant, and groups of 4 scenes per launch (24 frames per launch)
    for kw in (dict(pipelined=False), dict(pipelined=True, max_frames_per_launch=24), dict(pipelined=True, max_frames_per_launch=7)):
        for o in outs:
            o.zero_()
        assert render_clips(cms, "cama", outs, **kw) is True
        eng.join()
        torch.cuda.synchronize()
        bad = [k for k in range(24) if shard.overlay_hash(outs[k]) != golden[k]]
        assert not bad, (kw, bad)
    # scenes that do not agree (another image size) fall back to per-scene launches: same bytes, returns False
    b = _args(frames=6, verts=3000, height=90, width=160)
    other = bench.build_scene(b, 0, dev)[0]
    out_other = torch.zeros(eng.mosaic_shape(other._rig(), b.frames), dtype=torch.uint8, device=dev)
    outs[0].zero_()
    assert render_clips([cms[0], other], "cama", [outs[0], out_other], pipelined=True) is False
    eng.join()
    torch.cuda.synchronize()
    assert shard.overlay_hash(outs[0]) == golden[0]
    _, plain = other.render_clip("cama")
    torch.cuda.synchronize()
    assert torch.equal(plain, out_other)


def test_multi_scene_abi_rejects_bad_tables():
    """cama_render_scenes validates the HOST copy of the table before anything is enqueued."""
    import torch
    from cama_amd import _lib, runtime
    eng = runtime.engine()
    L = _lib.lib()
    host = np.zeros((2, 10), np.uint64)
    dev = torch.zeros((2, 10), dtype=torch.int64, device="cuda")
    crop = np.asarray(eng.crop, np.float64)
    scratch = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = L.cama_render_scenes(host.ctypes.data, dev.data_ptr(), 2, 0, dev.data_ptr(), 1, 6, crop.ctypes.data, 160, 96, 3, 2,
                              eng.halfwidth.ctypes.data, eng.palette.ctypes.data, scratch.data_ptr(), scratch.numel(), None)
    assert rc == -1 and b"calibration" in L.cama_last_error()
    rc = L.cama_render_scenes(None, dev.data_ptr(), 2, 0, dev.data_ptr(), 1, 6, crop.ctypes.data, 160, 96, 3, 2,
                              eng.halfwidth.ctypes.data, eng.palette.ctypes.data, scratch.data_ptr(), scratch.numel(), None)
    assert rc == -1 and b"scene table" in L.cama_last_error()
    rc = L.cama_render_scenes(host.ctypes.data, dev.data_ptr(), 70000, 0, dev.data_ptr(), 1, 6, crop.ctypes.data, 160, 96, 3, 2,
                              eng.halfwidth.ctypes.data, eng.palette.ctypes.data, scratch.data_ptr(), scratch.numel(), None)
    assert rc == -1 and b"S=70000" in L.cama_last_error()


def test_many_scenes_in_one_chain_with_spatially_sorted_maps():
    """Multi-scene chains on maps WITHOUT spatial order (random vertices: the engine renders from a Morton-sorted copy and
    carries the draw index per vertex as `draw_key`; the scene table then holds a key pointer instead of a colour pointer):
    same bytes as the per-scene launches, and the draw order (last writer wins) is the original one -- checked against the
    oracle on one scene."""
    import torch
    from cama_amd import runtime
    from cama_amd.dataset import render_clips
    a = _args(frames=4, verts=8000, height=180, width=320, map="random")
    dev = torch.device("cuda:0")
    scenes = [bench.build_scene(a, s, dev) for s in range(5)]
    cms = [cm for cm, _, _ in scenes]
    for cm in cms:
        d = cm._static("cama").device()
        assert d.sorted_soa is not None and d.sorted_key is not None and d.N == 8000
    eng = runtime.engine()
    shape = eng.mosaic_shape(cms[0]._rig(), a.frames)
    outs = [torch.zeros(shape, dtype=torch.uint8, device=dev) for _ in cms]
    assert render_clips(cms, "cama", outs, pipelined=True) is True
    eng.join()
    torch.cuda.synchronize()
    for k, cm in enumerate(cms):
        _, plain = cm.render_clip("cama")
        torch.cuda.synchronize()
        assert torch.equal(plain, outs[k]), k
    xyz, col, cams, w2c = G.scene_setup(a, 2)
    got = outs[2].cpu().numpy()
    for pos in range(a.frames):
        assert np.array_equal(got[pos], G.render_frame(a, 2, pos, xyz, col, cams, w2c)), pos


def test_two_threads_two_pipelines_share_the_mapping_table(sweep_scenes):
    """the overlay's per-buffer-pair choice of workgroup order is process-wide state behind a mutex.
    Two threads, each with its own Engine (own pipeline context, own streams) and its own full-size scene (2 GB read +
    2 GB written per launch, i.e. big enough for the tuner), render concurrently: 14 launches each, so both pairs go
    through their six trial launches and get decided while the other thread is launching too.  Every render must equal
    the oracle-checked golden hash; the table must end up with a decision for the last pair."""
    import threading
    import torch
    from cama_amd.engine import Engine
    a, scenes = sweep_scenes
    golden = _golden(a)
    dev = torch.device("cuda:0")
    errors, hashes = [], {}

    def worker(k):
        try:
            torch.cuda.set_device(dev)
            eng = Engine("cuda:0")
            cm, frames, _ = scenes[k]
            idx, w2c = cm.frame_poses("cama")
            rig, dmap = cm._rig(), cm._static("cama").device()
            out = torch.zeros(eng.mosaic_shape(rig, a.frames), dtype=torch.uint8, device=dev)
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                src = frames[1:1 + a.frames]
                got = []
                for rep in range(14):
                    out.fill_(0xA5)
                    eng.render_frames_pipelined(dmap, rig, w2c, src, out)
                    eng.join()
                    stream.synchronize()
                    got.append(shard.overlay_hash(out))
            hashes[k] = got
            hashes[("info", k)] = eng.overlay_mapping()
        except Exception as e:                                        # noqa: BLE001
            errors.append((k, repr(e)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors
    for k in (0, 1):
        assert all(h == golden[k] for h in hashes[k]), f"scene {k}: a concurrent render differs from the oracle's hash"
    last = [hashes[("info", k)] for k in (0, 1)]
    assert any(i["decided"] in (5, 31) and min(i["samples"]) >= 3 for i in last), last


@pytest.mark.gpu
def test_placed_buffers_render_the_same_bytes(monkeypatch):
    """Placement lives in the PRODUCT (round 5): ClipManager.render_clip(out=None) hands out a view of one of the engine's
    pooled buffers -- the fastest of CAMA_AUDITION candidate allocations for the clip's resident frames -- and moves the frames
    once into the fastest of a few candidates for that mosaic (DeviceFrameSource.place_for).  That changes WHERE the bytes
    live, never what they are; two successive ClipManagers of one shape reuse the pool (no new mosaic-sized allocation) and
    render the oracle's bytes; a ChunkedMosaic of pooled launches equals the plain render; the audition leaves the process
    options as they were."""
    import ctypes
    import gc
    import torch
    from cama_amd import runtime, _lib
    from cama_amd.engine import ChunkedMosaic
    a = _args()
    golden = _golden(a)
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    eng.pool.trim(0)
    monkeypatch.setenv("CAMA_AUDITION", "4")
    cm, frames, _ = bench.build_scene(a, 0, dev)
    rig = cm._rig()
    before = ctypes.c_int64(-7)
    _lib.check(eng.lib.cama_get_option(b"overlay_chunk_log2", ctypes.byref(before)))
    # the probe itself: a pure mosaic copy under its own kernel name, a time, and an error for what it cannot take
    probe_out = torch.empty(eng.mosaic_shape(rig, a.frames), dtype=torch.uint8, device=dev)
    ms = ctypes.c_double(0.0)
    _lib.check(eng.lib.cama_overlay_probe(frames[1:].data_ptr(), probe_out.data_ptr(), a.frames, rig.C, rig.H, rig.W, 3, 2,
                                          ctypes.byref(ms), eng._stream()))
    assert 0.05 < ms.value < 50.0
    want_copy = frames[1:1 + a.frames].reshape(a.frames, 2, 3, rig.H, rig.W, 3).permute(0, 1, 3, 2, 4, 5).reshape(probe_out.shape)
    assert torch.equal(probe_out, want_copy)
    assert eng.lib.cama_overlay_probe(frames[1:].data_ptr(), probe_out.data_ptr(), 1, rig.C, rig.H, 1592, 3, 1,
                                      ctypes.byref(ms), eng._stream()) != 0
    del probe_out, want_copy
    # first clip: the pool auditions 4 (.. 16) candidates for this source, the frames are placed against the winner
    n_log = len(getattr(eng, "audition_log", []))
    _, out = cm.render_clip("cama")
    torch.cuda.synchronize()
    logs = eng.audition_log[n_log:]
    mos = [e for e in logs if e["role"] == "mosaic"]
    frs = [e for e in logs if e["role"] == "frames"]
    # candidates are timed four at a time until one is fast or (from eight on) all are alike; CAMA_AUDITION=4: one group of four
    assert len(mos) == 1 and mos[0]["candidates"] == 4 and len(mos[0]["ms"]) == 4
    assert mos[0]["verdict"] in ("fast placement found", "budget") and mos[0]["peak_bytes"] == 4 * out.numel()
    assert mos[0]["chosen_ms"] == min(mos[0]["ms"]) and mos[0]["source"] == "engine pool" and mos[0]["seconds"] > 0
    assert len(frs) == 1 and len(frs[0]["ms"]) == 3                        # the caller's tensor + 2 candidates (CAMA_AUDITION // 2)
    assert eng.pool.stats["audition_peak_bytes"] >= 4 * out.numel() and eng.pool.stats["audition_seconds"] > 0
    assert torch.equal(cm.frame_source().frames, frames)                   # moved or not: the same bytes
    assert shard.overlay_hash(out) == golden[0]
    base_ptr = out.data_ptr()
    stats0 = dict(eng.pool.stats)
    # the pipelined path into the same pooled buffer
    out.fill_(0xA5)
    cm.render_clip("cama", out=out, pipelined=True)
    eng.join()
    torch.cuda.synchronize()
    assert shard.overlay_hash(out) == golden[0]
    after = ctypes.c_int64(-7)
    _lib.check(eng.lib.cama_get_option(b"overlay_chunk_log2", ctypes.byref(after)))
    assert after.value == before.value
    # second ClipManager of the same shape: the pool serves it -- no audition, no new mosaic-sized allocation
    del out, cm
    gc.collect()
    mem0 = torch.cuda.memory_allocated(dev)
    cm2, frames2, _ = bench.build_scene(a, 1, dev)
    mem1 = torch.cuda.memory_allocated(dev)
    _, out2 = cm2.render_clip("cama")
    torch.cuda.synchronize()
    assert out2.data_ptr() == base_ptr
    assert eng.pool.stats["hits"] == stats0["hits"] + 1 and eng.pool.stats["auditions"] == stats0["auditions"]
    assert eng.pool.stats["allocations"] == stats0["allocations"]
    mosaic_bytes = out2.numel()
    # (what the second clip added beyond its own frames + its placed copy of them: scratch and poses, far below one mosaic)
    grown = torch.cuda.memory_allocated(dev) - mem1
    assert grown < frames2.numel() + mosaic_bytes // 2, (grown, frames2.numel(), mosaic_bytes)
    assert shard.overlay_hash(out2) == golden[1]
    # chunked: launches of 24 + 16 frames, each into its own pooled allocation (dataset.MOSAIC_CHUNK_BYTES below the clip's size)
    from cama_amd import dataset as cama_dataset
    monkeypatch.setattr(cama_dataset, "MOSAIC_CHUNK_BYTES", 1 << 29)
    _, chunked = cm2.render_clip("cama", frames_per_launch=24, pipelined=True)
    eng.join()
    torch.cuda.synchronize()
    assert isinstance(chunked, ChunkedMosaic) and chunked.shape == tuple(out2.shape) and chunked.bounds == [0, 24, 40]
    assert torch.equal(torch.cat([c for _, _, c in chunked.spans()]), out2)
    monkeypatch.setenv("CAMA_AUDITION", "0")                               # no audition: plain allocations, still pooled
    del chunked
    _, chunked = cm2.render_clip("cama", frames_per_launch=24)
    assert isinstance(chunked, ChunkedMosaic)
    del chunked, out2
    eng.pool.trim(0)


@pytest.mark.gpu
def test_memoised_launch_list_replays_the_same_bytes_and_notices_every_change():
    """Round 5: a pipelined render_clip into the same buffers works out its launches once (Engine.clip_desc + one
    cama_pipeline_render_clip call per launch) and replays them while everything they were derived from is still the very
    same object.  The replay writes the bytes the un-memoised path writes (= the oracle's, golden hashes); a new mosaic, a new
    frame tensor, a changed crop box, a re-parsed pose track or dataset.LAUNCH_MEMO = False each send the next call down the full path."""
    import torch
    from cama_amd import runtime
    from cama_amd.frames import DeviceFrameSource
    a = _args(frames=12, height=270, width=480)
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    cm, frames, _ = bench.build_scene(a, 3, dev)
    _, plain = cm.render_clip("cama")                                  # single stream, no memo
    want = plain.clone()
    out = torch.empty_like(want)
    calls = []
    real = eng.render_clip_launch
    eng.render_clip_launch = lambda *args: (calls.append(args[2]), real(*args))[1]
    try:
        def step(o=out, **kw):
            o.fill_(0xA5)
            cm.render_clip("cama", out=o, pipelined=True, **kw)
            eng.join()
            torch.cuda.synchronize()
            assert torch.equal(o, want)
        step()
        memo = cm._launch_memo["cama"]
        assert len(memo["launches"]) == 1 and calls == [12]
        step()
        assert cm._launch_memo["cama"] is memo and calls == [12, 12]          # replayed
        step(frames_per_launch=5)                                             # another launch size: planned again (5 + 5 + 2)
        assert cm._launch_memo["cama"] is not memo and calls[2:] == [5, 5, 2]
        memo = cm._launch_memo["cama"]
        step(frames_per_launch=5)
        assert cm._launch_memo["cama"] is memo
        # out of memory inside the replay (the library could not grow its scratch): the launches already issued stay, the rest
        # goes through the general path, the bytes are the same and the memo is gone
        n_before = len(calls)
        ok_launch = eng.render_clip_launch

        def failing(*args):
            if len(calls) - n_before == 1:
                eng.render_clip_launch = ok_launch
                raise torch.OutOfMemoryError("simulated CAMA_ENOMEM")
            return ok_launch(*args)
        eng.render_clip_launch = failing
        step(frames_per_launch=5)
        eng.render_clip_launch = ok_launch
        assert calls[n_before:n_before + 1] == [5]
        step(frames_per_launch=5)                                             # planned again, replayable again
        memo = cm._launch_memo["cama"]
        other = torch.empty_like(want)
        step(other, frames_per_launch=5)                                      # another mosaic
        assert cm._launch_memo["cama"]["out"] is other
        memo = cm._launch_memo["cama"]
        cm.set_frame_source(DeviceFrameSource(frames.clone()))                # another source object
        step(other, frames_per_launch=5)
        assert cm._launch_memo["cama"] is not memo
        memo = cm._launch_memo["cama"]
        cm._track_cache.pop("cama")                                           # re-parsed track: new pose arrays
        step(other, frames_per_launch=5)
        assert cm._launch_memo["cama"] is not memo
        # a changed crop box changes the picture: the memo must not replay the old one
        cm.mm.crop_dict["x_max"] = 5.0
        _, cropped = cm.render_clip("cama")
        assert not torch.equal(cropped, want)
        other.fill_(0xA5)
        cm.render_clip("cama", out=other, pipelined=True, frames_per_launch=5)
        eng.join()
        torch.cuda.synchronize()
        assert torch.equal(other, cropped)
    finally:
        eng.render_clip_launch = real
    # raw sensor frames through the 3:5 kernel: the same one-call form (kind 1)
    b = _args(frames=6, height=540, width=960, raw_frames=True)
    cm2, raw, _ = bench.build_scene(b, 4, dev)
    _, plain2 = cm2.render_clip("cama")
    o2 = torch.full_like(plain2, 0xA5)
    for _ in range(2):
        cm2.render_clip("cama", out=o2, pipelined=True)
    eng.join()
    torch.cuda.synchronize()
    assert torch.equal(o2, plain2) and cm2._launch_memo["cama"]["desc"].kind == 1


@pytest.mark.gpu
def test_band_height_is_chosen_per_launch_and_never_changes_a_byte():
    """8-row bands pay on maps that stamp the image densely and cost 2-4 % elsewhere, so a pipeline picks the
    height per launch from what the same map produced before (cama_pipeline BandMemo; first launch over a map: 4 rows).  Here:
    both heights FORCED on one dense scene give the plain path's bytes; left to itself the pipeline moves a dense map to 8 rows
    after its first launches and leaves a 10^4-vertex map at 4 -- same bytes throughout."""
    import ctypes
    import torch
    from cama_amd import runtime, _lib
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    L = eng.lib

    def info():
        return eng.pipeline_info() or {"tall_band_launches": 0, "last_band_rows": 0}

    def set_rows(v):
        _lib.check(L.cama_set_option(b"band_rows", v))

    a = _args(verts=1000000, frames=6)
    cm, frames, _ = bench.build_scene(a, 0, dev)
    _, plain = cm.render_clip("cama")                                  # single stream: band_rows_for(W) = 4
    want = plain.clone()
    out = torch.empty_like(want)

    def step(cmx=cm, o=out, w=want):
        o.fill_(0xA5)
        cmx.render_clip("cama", out=o, pipelined=True)
        eng.join()
        torch.cuda.synchronize()
        assert torch.equal(o, w)
    try:
        for rows in (8, 4, 8):
            set_rows(rows)
            n0 = info()["tall_band_launches"]
            step()
            assert info()["last_band_rows"] == rows and info()["tall_band_launches"] - n0 == (1 if rows == 8 else 0)
        set_rows(0)                                                    # the pipeline's own choice
        n0 = info()["tall_band_launches"]
        for _ in range(6):                                             # the read-back of launch k is in hand two launches later at most
            step()
        assert info()["tall_band_launches"] > n0 and info()["last_band_rows"] == 8
        # a clip-sized map never moves (and never pays for the read-back: N < 200 000)
        b = _args(frames=6)
        cm2, frames2, _ = bench.build_scene(b, 1, dev)
        _, plain2 = cm2.render_clip("cama")
        want2, out2 = plain2.clone(), torch.empty_like(plain2)
        n0 = info()["tall_band_launches"]
        for _ in range(4):
            step(cm2, out2, want2)
        assert info()["tall_band_launches"] == n0 and info()["last_band_rows"] == 4
        # a height the launch cannot take (owner table beyond half the LDS, or below 2 x radius) falls back to the default
        set_rows(16)
        step(cm2, out2, want2)
        assert info()["last_band_rows"] in (4, 16)
    finally:
        set_rows(0)


@pytest.mark.gpu
def test_fullsize_site_scenes_equal_the_oracle():
    """configs[3] at FULL size as bench.py runs it on one GPU (12 scenes over three 10^6-vertex site maps, 40 frames at
    1600x900): one scene of every site plus a second drive over scene 0's site,
    rendered through the planned pipeline (cull pre-pass on its own stream, demand-sized scratch) -- whole-scene hashes against
    the oracle-rendered golden hashes (tests/golden/gen_scene_hashes.py site-full)."""
    import torch
    from cama_amd import runtime
    a = _args(map="site", verts=1000000, sites=3, scenes=12)
    golden = shard.load_golden_hashes(GOLDEN, bench.args_key(a))
    assert golden and len(golden) == 12
    dev = torch.device("cuda:0")
    eng = runtime.engine()
    out = None
    for sid in (0, 1, 2, 3):
        cm, frames, _ = bench.build_scene(a, sid, dev)
        if out is None:
            out = torch.empty(eng.mosaic_shape(cm._rig(), a.frames), dtype=torch.uint8, device=dev)
        out.fill_(0xA5)
        cm.render_clip("cama", out=out, pipelined=True)
        eng.join()
        torch.cuda.synchronize()
        assert shard.overlay_hash(out) == golden[sid], f"scene {sid} (site {sid % 3})"
        del cm, frames
    assert eng.pipeline_info()["planned_launches"] >= 4
