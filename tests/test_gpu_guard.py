"""Out-of-bounds hunting for the fused render chain ("a fault whose cause was not found in a chain that passes
kernel by kernel is a latent-bug smell -- a real out-of-bounds hidden by allocator slack").

Every buffer the chain touches is caller-owned, and torch's caching allocator hands out blocks with slack around them, so an
indexing error by a few elements would normally go unnoticed.  Here the chain (memset, block cameras, projection, scans,
scatter, overlay -- plain, work-list, multi-scene, raw 3:5) runs on buffers with NO slack:

  * writes: the scratch is exactly cama_render_scratch_bytes() long inside a larger tensor whose bytes before and after it
    carry a pattern; after the run the pattern must be intact (the same for the mosaic);
  * reads: vertex buffer, frames and scratch each END exactly at the end of their own hipMalloc allocation of whole 2 MB
    pages (raw hipMalloc through the runtime torch loaded, not the caching allocator): a read or write past the end lands
    on the next, unmapped page and kills the process with a GPU memory fault.  That part runs in a subprocess.

The rendered bytes are compared with the engine's ordinary path, so the guarded runs are also parity runs."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GUARD = 1 << 20
PATTERN = 0xA5


def _scene(seed, N, F, W, H, spread=60.0):
    from tests.test_gpu_kernels import _random_scene
    return _random_scene(seed, N, F, W, H, spread=spread)


def _render_with_guarded_scratch(eng, dmap, rig, w2c, src, crop=None):
    """cama_render_frames with the scratch exactly sized between two pattern-filled guard zones and the mosaic between
    two more; returns the mosaic.  Asserts the guards afterwards."""
    import torch
    from cama_amd import _lib
    L = eng.lib
    F = len(w2c)
    cropa = eng._crop(crop)
    need = int(L.cama_render_scratch_bytes(dmap.N, F, rig.C, rig.H, rig.W, eng.radius))
    need16 = (need + 255) // 256 * 256
    buf = torch.full((GUARD + need16 + GUARD,), PATTERN, dtype=torch.uint8, device=eng.device)
    scratch = buf[GUARD:GUARD + need]
    shape = eng.mosaic_shape(rig, F)
    mbytes = int(np.prod(shape))
    mbuf = torch.full((GUARD + mbytes + GUARD,), PATTERN, dtype=torch.uint8, device=eng.device)
    out = mbuf[GUARD:GUARD + mbytes].view(shape)
    T = eng._mats(w2c)
    x, y, z, col, key, bnd, flags = dmap.render_ptrs(cropa)
    _lib.check(L.cama_render_frames(x, y, z, dmap.is_f64, col, key, bnd, flags, dmap.N, T.data_ptr(), F, rig.c2cam.data_ptr(),
                                    rig.K.data_ptr(), rig.C, cropa.ctypes.data, rig.W, rig.H, src.data_ptr(), out.data_ptr(), 3,
                                    eng.radius, eng.halfwidth.ctypes.data, eng.palette.ctypes.data, scratch.data_ptr(), need,
                                    eng._stream()))
    torch.cuda.synchronize()
    for name, b, n in (("scratch", buf, need16), ("mosaic", mbuf, mbytes)):
        assert bool((b[:GUARD] == PATTERN).all()), f"{name}: bytes BEFORE the buffer were written"
        tail = b[GUARD + (need if name == "scratch" else n):]
        assert bool((tail == PATTERN).all()), f"{name}: bytes AFTER the buffer were written"
    return out.clone()


@pytest.mark.parametrize("N,F,W,H,spread,force_bounds", [
    (3000, 3, 160, 96, 60.0, False),          # small map, plain grid
    (9996, 5, 960, 540, 60.0, False),         # the headline's map size, vector path
    (257, 2, 100, 37, 60.0, False),           # generic-width overlay, ragged vertex block
    (70000, 4, 320, 180, 60.0, True),         # block index: camera masks, several vertex blocks per workgroup
    (70000, 4, 320, 180, 300.0, True),        # site-sized: work lists + persistent workgroups
    (1, 1, 64, 32, 5.0, False),               # one vertex
])
def test_chain_writes_stay_inside_exactly_sized_buffers(N, F, W, H, spread, force_bounds, monkeypatch):
    import torch
    from cama_amd import engine as E
    if force_bounds:
        monkeypatch.setattr(E, "BOUNDS_MIN_VERTS", 1)
    eng = E.Engine("cuda:0")
    xyz, col, cams, w2c = _scene(7, N, F, W, H, spread)
    rig = eng.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
    dmap = eng.upload_map(xyz, col)
    src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    want = eng.render_frames(dmap, rig, w2c, src).clone()
    got = _render_with_guarded_scratch(eng, dmap, rig, w2c, src)
    assert torch.equal(got, want)


_CHILD = r"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["CAMA_REPO"])
from cama_amd import _lib, engine as E
from tests.test_gpu_kernels import _random_scene

hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name)          # the runtime torch already loaded (same process-wide instance)
        break
    except OSError:
        pass
assert hip is not None
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
PAGE = 2 << 20


def at_end_of_pages(nbytes, align=256):
    # device pointer p with [p, p + nbytes) ending exactly at the end of a hipMalloc of whole 2 MB pages
    total = (nbytes + align + PAGE - 1) // PAGE * PAGE
    base = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(base), total) == 0
    assert base.value % PAGE == 0, "hipMalloc did not return a 2 MB aligned block"
    assert hip.hipMemset(base, 0x5A, total) == 0
    p = base.value + total - nbytes
    assert p % 16 == 0 or nbytes % 16, (p, nbytes)
    return p


def put(arr):
    a = np.ascontiguousarray(arr)
    n = a.nbytes
    pad = (-n) % 16
    p = at_end_of_pages(n + pad) + pad          # the array itself ends at the page end
    assert hip.hipMemcpy(p, a.ctypes.data, n, 1) == 0
    return p


mode = sys.argv[1]
N, F, W, H, spread, bounds = {"plain": (9996, 3, 320, 180, 60.0, False), "bounds": (70000, 3, 320, 180, 60.0, True),
                              "site": (70000, 3, 320, 180, 300.0, True),
                              "control": (9996, 3, 320, 180, 60.0, False)}[mode]
LIE = 8192 if mode == "control" else 0        # positive control: claim more vertices than the buffers hold
if bounds:
    E.BOUNDS_MIN_VERTS = 1
eng = E.Engine("cuda:0")
L = eng.lib
xyz, col, cams, w2c = _random_scene(11, N, F, W, H, spread=spread)
rig = eng.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
dmap = eng.upload_map(xyz, col)                                  # (spatial index etc. from the ordinary path)
src_t = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
want = eng.render_frames(dmap, rig, w2c, src_t).clone()
torch.cuda.synchronize()
# the same inputs again, each at the very end of its own allocation
soa = (dmap.sorted_soa if dmap.sorted_soa is not None else dmap.soa).cpu().numpy()
px, py, pz = put(soa[0]), put(soa[1]), put(soa[2])
pcol = put(dmap.colour.cpu().numpy())
pkey = put(dmap.sorted_key.cpu().numpy()) if dmap.sorted_key is not None else None
cropa = eng._crop(None)
_, _, _, _, _, bnd, flags = dmap.render_ptrs(cropa)
pbnd = put(dmap.bounds.cpu().numpy()) if bnd is not None else None
psrc = put(src_t.cpu().numpy())
pT = put(np.asarray(w2c, np.float64).reshape(F, 16))
pc2c, pK = put(rig.c2cam_host), put(rig.K_host)
need = int(L.cama_render_scratch_bytes(dmap.N + LIE, F, rig.C, rig.H, rig.W, eng.radius))
pscr = at_end_of_pages(need)
shape = eng.mosaic_shape(rig, F)
nout = int(np.prod(shape))
pout = at_end_of_pages(nout)
_lib.check(L.cama_render_frames(px, py, pz, dmap.is_f64, pcol, pkey, pbnd, flags, dmap.N + LIE, pT, F, pc2c, pK, rig.C,
                                cropa.ctypes.data, rig.W, rig.H, psrc, pout, 3, eng.radius, eng.halfwidth.ctypes.data,
                                eng.palette.ctypes.data, pscr, need, None))
torch.cuda.synchronize()
got = np.empty(nout, np.uint8)
assert hip.hipMemcpy(got.ctypes.data, pout, nout, 2) == 0
assert np.array_equal(got.reshape(shape), want.cpu().numpy()), "guard-paged render differs"
print("GUARD_OK", mode, N, flags)
"""


@pytest.mark.parametrize("mode", ["plain", "bounds", "site"])
def test_chain_does_not_touch_the_page_after_any_buffer(mode, repo_root, tmp_path):
    script = tmp_path / "guard_child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, CAMA_REPO=repo_root, PYTHONPATH=repo_root)
    p = subprocess.run([sys.executable, str(script), mode], cwd=repo_root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "GUARD_OK" in p.stdout, (p.returncode, p.stdout[-1500:], p.stderr[-3000:])


def test_guard_pages_do_catch_an_overrun(repo_root, tmp_path):
    """Positive control for the test above: the same child, told that the map has 8 192 more vertices than its buffers
    hold -- the projection then reads 32 KB past the end of x / y / z, onto the next page.  The child must die (GPU memory
    fault) or at least not print GUARD_OK; if it does, guard pages are not effective on this box and the test above proves
    less than it says."""
    script = tmp_path / "guard_child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, CAMA_REPO=repo_root, PYTHONPATH=repo_root)
    p = subprocess.run([sys.executable, str(script), "control"], cwd=repo_root, env=env, capture_output=True, text=True,
                       timeout=600)
    if p.returncode == 0 and "GUARD_OK" in p.stdout:
        pytest.xfail("the page after a 2 MB-granular hipMalloc is mapped on this box: overruns would go unnoticed")
    assert p.returncode != 0


def test_planned_launches_size_their_scratch_from_the_cull_and_stay_inside_it(monkeypatch):
    """site-sized maps no longer get 24 B of stamp scratch per (frame, camera, vertex).  The pipeline runs
    the cull pre-pass first, reads back what survives -- blocks of the busiest frame, (wave, camera) chains -- and sizes its
    own stamp buffers from that exact bound (cama_pipeline_render with scratch0 == NULL).  Checked here: the planned render
    equals the worst-case-sized one (caller scratch, exactly sized between guard zones) byte for byte; the pipeline's buffers
    are a small fraction of the worst case; their own guard zones are intact after launches of growing and shrinking
    demand; a launch with NO survivor at all and one where every block survives both work."""
    import torch
    from cama_amd import engine as E
    monkeypatch.setattr(E, "BOUNDS_MIN_VERTS", 1)
    eng = E.Engine("cuda:0")
    assert eng.lib.cama_set_option(b"cull_list_min", 1) == 0     # (the work-list route normally starts at 16 k items)
    try:
        _planned_launches(eng)
    finally:
        eng.lib.cama_set_option(b"cull_list_min", 16384)


def _planned_launches(eng):
    import torch
    N, W, H = 120000, 320, 180
    outs = {}
    worst = {}
    for tag, F, spread, seed in (("site", 12, 300.0, 3), ("site-long", 40, 300.0, 4), ("site-short", 2, 300.0, 5),
                                 ("dense", 6, 40.0, 6)):
        xyz, col, cams, w2c = _scene(seed, N, F, W, H, spread)
        rig = eng.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
        dmap = eng.upload_map(xyz, col)
        src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
        ptrs = dmap.render_ptrs(eng.crop)
        want = _render_with_guarded_scratch(eng, dmap, rig, w2c, src)              # worst case, caller scratch, guarded
        out = torch.zeros_like(want)
        eng.render_frames_pipelined(dmap, rig, np.asarray(w2c, np.float32), src, out)
        eng.join()
        torch.cuda.synchronize()
        assert torch.equal(out, want), tag
        info = eng.pipeline_info()
        worst[tag] = int(eng.lib.cama_render_scratch_bytes(N, F, rig.C, H, W, eng.radius))
        outs[tag] = (info, bool(ptrs[6] & 1))
        bad = np.zeros(1, np.int64)
        from cama_amd import _lib
        _lib.check(eng.lib.cama_pipeline_guard_check(eng._pipe["handle"], bad.ctypes.data))
        assert int(bad[0]) == 0, (tag, int(bad[0]))
    # site-sized maps were planned, the clip-sized one (every block in view) was not
    assert outs["site"][1] and outs["site"][0]["planned_launches"] >= 1 and outs["site"][0]["last_plan_capacity"] > 0
    assert outs["site-long"][0]["planned_launches"] > outs["site"][0]["planned_launches"]
    assert not outs["dense"][1] and outs["dense"][0]["planned_launches"] == outs["site-short"][0]["planned_launches"]
    # two slots of demand-sized scratch against ONE worst-case buffer: well below a tenth for the 40-frame site launch
    assert outs["site-long"][0]["scratch_bytes"] < 0.1 * worst["site-long"], (outs["site-long"][0], worst["site-long"])
    # a frame batch with no survivor anywhere (the car far outside the map): still a pure mosaic copy
    xyz, col, cams, w2c = _scene(9, N, 3, W, H, 300.0)
    far = np.asarray(w2c, np.float32).copy()
    far[:, 0, 3] += 5000.0
    rig = eng.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
    dmap = eng.upload_map(xyz, col)
    src = torch.randint(0, 256, (3, 6, H, W, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros(eng.mosaic_shape(rig, 3), dtype=torch.uint8, device="cuda")
    eng.render_frames_pipelined(dmap, rig, far, src, out)
    eng.join()
    torch.cuda.synchronize()
    want = torch.cat([torch.cat([src[:, c] for c in range(r * 3, r * 3 + 3)], dim=2) for r in range(2)], dim=1)
    assert torch.equal(out, want)
    assert eng.pipeline_info()["last_plan_capacity"] >= 1


def test_planned_and_unplanned_launches_interleaved_without_joins(monkeypatch):
    """Round 4: the cull pre-pass and the pose upload of PLANNED launches run on the pipeline's third stream, ordered by
    events behind the chain that last used the slot; whether a launch's staged poses went up on that stream or on the
    binning stream depends on what the PREVIOUS launch was.  Twenty launches back to back on one pipeline -- planned (site map)
    and unplanned (dense map) in every order of succession, staged host poses, different frame counts, no join in between --
    must each equal their plain single-stream render."""
    import torch
    from cama_amd import engine as E
    monkeypatch.setattr(E, "BOUNDS_MIN_VERTS", 1)
    eng = E.Engine("cuda:0")
    assert eng.lib.cama_set_option(b"cull_list_min", 1) == 0
    try:
        N, W, H = 60000, 320, 180
        scenes = []
        for tag, F, spread, seed in (("site", 9, 300.0, 11), ("dense", 5, 40.0, 12), ("site2", 14, 300.0, 13), ("dense2", 3, 40.0, 14)):
            xyz, col, cams, w2c = _scene(seed, N, F, W, H, spread)
            rig = eng.make_rig([c["name"] for c in cams], [c["chassis2camera"] for c in cams], [c["K"] for c in cams], W, H)
            dmap = eng.upload_map(xyz, col)
            src = torch.randint(0, 256, (F, 6, H, W, 3), dtype=torch.uint8, device="cuda")
            want = eng.render_frames(dmap, rig, w2c, src).clone()
            scenes.append((tag, dmap, rig, np.asarray(w2c, np.float32), src, want))
        torch.cuda.synchronize()
        before = eng.pipeline_info()["planned_launches"] if getattr(eng, "_pipe", None) else 0
        order = [0, 0, 1, 1, 0, 2, 1, 3, 2, 2, 3, 0, 3, 1, 2, 0, 1, 0, 3, 2]      # planned->planned, ->unplanned, and back
        outs = []
        for k in order:
            tag, dmap, rig, w2c, src, want = scenes[k]
            out = torch.full_like(want, 0xA5)
            eng.render_frames_pipelined(dmap, rig, w2c, src, out)
            outs.append((k, out))
        eng.join()
        torch.cuda.synchronize()
        for n, (k, out) in enumerate(outs):
            assert torch.equal(out, scenes[k][5]), (n, scenes[k][0])
        info = eng.pipeline_info()
        assert info["planned_launches"] - before == sum(1 for k in order if k in (0, 2))
    finally:
        eng.lib.cama_set_option(b"cull_list_min", 16384)
