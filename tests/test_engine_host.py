"""Host-side pieces of cama_amd.engine that need no GPU."""
import builtins
import sys

import numpy as np

from cama_amd import engine


def test_content_hasher_falls_back_to_hashlib_without_xxhash(monkeypatch):
    """Engine.shared_map keys device maps on their content; xxhash is optional (requirements.txt) -- without it the key
    comes from hashlib.blake2b, still 128 bits and still a function of the bytes only."""
    data = np.arange(1000, dtype=np.float32).view(np.uint8).data
    real_import = builtins.__import__

    def no_xxhash(name, *a, **k):
        if name == "xxhash":
            raise ImportError("blocked for the test")
        return real_import(name, *a, **k)

    monkeypatch.delitem(sys.modules, "xxhash", raising=False)
    monkeypatch.setattr(builtins, "__import__", no_xxhash)
    h1, h2 = engine._content_hasher(), engine._content_hasher()
    assert type(h1).__module__.startswith("_blake2") or "blake2" in type(h1).__name__.lower()
    h1.update(data)
    h2.update(data)
    assert h1.hexdigest() == h2.hexdigest() and len(h1.hexdigest()) == 32
    h3 = engine._content_hasher()
    h3.update(np.arange(1, 1001, dtype=np.float32).view(np.uint8).data)
    assert h3.hexdigest() != h1.hexdigest()


def test_requirements_file_lists_what_the_default_path_imports():
    import os
    req = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "requirements.txt")).read()
    for name in ("numpy", "scipy", "PyYAML", "tqdm", "torch", "Pillow"):
        assert name in req


def test_chunked_mosaic_indexes_frames_and_refuses_slices_across_chunks():
    """engine.ChunkedMosaic: a long clip's mosaic as one allocation per launch (MosaicPool.take_many).  Frames by int,
    views by slices inside one chunk; ClipManager.render_clip cuts its launches at `bounds`."""
    import pytest
    import torch
    chunks = [torch.full((n, 2, 4, 3), k, dtype=torch.uint8) for k, n in enumerate((4, 4, 3))]
    m = engine.ChunkedMosaic(chunks)
    assert m.shape == (11, 2, 4, 3) and len(m) == 11 and m.bounds == [0, 4, 8, 11]
    assert [int(m[f][0, 0, 0]) for f in range(11)] == [0] * 4 + [1] * 4 + [2] * 3 and int(m[-1][0, 0, 0]) == 2
    v = m[4:8]
    assert v.shape[0] == 4 and v.data_ptr() == chunks[1].data_ptr()
    assert m[9:11].data_ptr() == chunks[2][1:].data_ptr() and m[:3].shape[0] == 3 and m[5:5].shape[0] == 0
    with pytest.raises(IndexError):
        m[3:5]
    with pytest.raises(IndexError):
        m[11]
    with pytest.raises(IndexError):
        m[0:8:2]
    m.fill_(7)
    assert all(int(c.min()) == 7 and int(c.max()) == 7 for c in chunks)
    assert [(lo, hi) for lo, hi, _ in m.spans()] == [(0, 4), (4, 8), (8, 11)]


def test_placement_helpers_only_probe_what_the_probe_kernel_can_read():
    """MosaicPool.take / take_many and Engine.place_frames fall back to plain allocations unless the source is a contiguous
    uint8 [F, C, H, W, 3] tensor of the rig's size with W % 16 == 0 and opaque stamps (what cama_overlay_probe takes)."""
    import types
    import torch
    eng = types.SimpleNamespace(alpha256=256)
    rig = types.SimpleNamespace(C=6, H=4, W=32)
    ok = torch.zeros((3, 6, 4, 32, 3), dtype=torch.uint8)
    probe = engine.Engine._probeable
    assert probe(eng, rig, ok)
    assert probe(eng, rig, ok[1:])                                   # a slice along frames stays contiguous
    assert not probe(eng, rig, ok[:, :, :, ::2])                     # strided view
    assert not probe(eng, rig, ok.permute(0, 1, 3, 2, 4))            # wrong layout
    assert not probe(eng, rig, ok.to(torch.int8))
    assert not probe(eng, types.SimpleNamespace(C=6, H=4, W=24), torch.zeros((3, 6, 4, 24, 3), dtype=torch.uint8))   # W % 16
    assert not probe(types.SimpleNamespace(alpha256=128), rig, ok)   # translucent stamps go through another kernel
    assert not probe(eng, rig, torch.zeros((3, 6, 4, 32), dtype=torch.uint8))


def _cpu_pool():
    """A MosaicPool over CPU tensors: the bookkeeping (who still references a base, which base serves which request) is
    device-independent; only the audition needs a GPU."""
    import types
    import torch
    eng = types.SimpleNamespace(device=torch.device("cpu"), alpha256=256, _log_audition=lambda e: None)
    return engine.MosaicPool(eng)


def test_mosaic_pool_recycles_a_base_only_when_no_view_of_it_is_left():
    """the mosaics the product allocates for a caller are views of long-lived, engine-owned buffers.  A
    base is lent again only when NOTHING but the pool references its storage -- a frame sliced out of a mosaic the caller has
    already dropped keeps the whole base out of circulation (the reference's caller may hold a frame as long as it likes)."""
    pool = _cpu_pool()
    a = pool.take((4, 6, 8, 3))
    assert tuple(a.shape) == (4, 6, 8, 3) and pool.stats["allocations"] == 1 and pool.stats["hits"] == 0
    ptr = a.data_ptr()
    b = pool.take((4, 6, 8, 3))                       # `a` is still out: a second base
    assert b.data_ptr() != ptr and pool.stats["allocations"] == 2
    frame = a[2]                                      # a view of a view ...
    del a
    c = pool.take((4, 6, 8, 3))                       # ... still pins the base
    assert c.data_ptr() not in (ptr, b.data_ptr()) and pool.stats["allocations"] == 3
    frame.fill_(9)
    del frame
    d = pool.take((4, 6, 8, 3))                       # now it is idle: recycled, no allocation
    assert d.data_ptr() == ptr and pool.stats["allocations"] == 3 and pool.stats["hits"] == 1
    assert len(pool.bases) == 3 and pool.nbytes() == 3 * 4 * 6 * 8 * 3


def test_mosaic_pool_serves_shorter_requests_from_a_longer_idle_base_and_trims():
    pool = _cpu_pool()
    a = pool.take((16, 2, 4, 3))
    ptr = a.data_ptr()
    del a
    b = pool.take((12, 2, 4, 3))                      # 12 <= 16 <= 12 + 8: the idle 16-frame base serves it
    assert b.data_ptr() == ptr and tuple(b.shape) == (12, 2, 4, 3) and pool.stats["hits"] == 1
    del b
    c = pool.take((2, 2, 4, 3))                       # 16 > 2 + 8: too much would lie fallow -> its own base
    assert c.data_ptr() != ptr and pool.stats["allocations"] == 2
    d = pool.take((16, 2, 8, 3))                      # another frame shape never matches
    assert pool.stats["allocations"] == 3
    assert pool.trim(0) == 1 and len(pool.bases) == 2  # only the idle base goes; lent ones stay
    del c, d
    assert pool.trim(0) == 2 and pool.bases == [] and pool.stats["trimmed"] == 3


def test_mosaic_pool_take_many_uses_idle_bases_first_and_never_lends_one_base_twice():
    pool = _cpu_pool()
    first = pool.take_many([(5, 2, 4, 3)] * 3, None, [None] * 3)
    ptrs = sorted(t.data_ptr() for t in first)
    assert len(set(ptrs)) == 3 and pool.stats["allocations"] == 3
    del first
    again = pool.take_many([(5, 2, 4, 3)] * 4, None, [None] * 4)
    got = sorted(t.data_ptr() for t in again)
    assert len(set(got)) == 4 and set(ptrs) <= set(got)          # the three idle bases + one new
    assert pool.stats["hits"] == 3 and pool.stats["allocations"] == 4


def test_audition_stops_as_soon_as_it_knows():
    """on a box without a fast placement (round 5, the driver's box: 72 candidates within 2.1 %) the audition
    must cost eight candidates, not 72; a fast candidate ends it at once; otherwise CAMA_AUDITION bounds it."""
    shape = (2, 4, 8, 3)
    nbytes = 2 * 4 * 8 * 3
    good = 2.0 * nbytes / (engine.MosaicPool.GOOD_FRAC * 8.0e12) * 1e3

    def run(seq, K=16, n_keep=1):
        pool = _cpu_pool()
        it = iter(seq)
        cands, times, verdict, secs, peak = pool._audition(shape, n_keep, lambda c: next(it), K)
        assert len(cands) == len(times) and peak == len(cands) * nbytes and secs >= 0
        assert len({c.data_ptr() for c in cands}) == len(cands)              # alive together: distinct memory
        return pool, times, verdict

    flat = [good * 1.06 * (1 + 0.002 * (k % 5)) for k in range(64)]           # all slow, within 1 %
    pool, times, verdict = run(flat)
    assert len(times) == 8 and verdict == "no fast mode on this box" and pool.flat_box
    fast_third = [good * 1.06, good * 1.07, good * 0.99, good * 1.05] + flat
    pool, times, verdict = run(fast_third)
    assert len(times) == 4 and verdict == "fast placement found" and not pool.flat_box
    spread = [good * (1.03 + 0.006 * (k % 9)) for k in range(64)]               # none fast, but 4 % apart: not flat either
    pool, times, verdict = run(spread)
    assert len(times) == 16 and verdict == "budget" and not pool.flat_box
    pool, times, verdict = run(spread, K=6)
    assert len(times) == 6 and verdict == "budget"
    # several buffers at once (take_many): it goes on until n_keep fast ones are in hand
    two_fast = [good * 1.06] * 3 + [good * 0.99] + [good * 1.06] * 3 + [good * 0.98] + flat
    pool, times, verdict = run(two_fast, K=32, n_keep=2)
    assert len(times) == 8 and verdict == "fast placement found"
    # a flat box is remembered: the next take() does not audition (no rig / source needed to see that it allocates plainly)
    pool, _, _ = run(flat)
    t = pool.take(shape)
    assert pool.stats["auditions"] == 0 and pool.stats["allocations"] == 1 and tuple(t.shape) == shape


def test_replaying_a_memoised_launch_list_survives_out_of_memory():
    """a CAMA_ENOMEM from the library inside the memoised replay (torch.OutOfMemoryError) must not escape
    render_clip: the memo is dropped, the engine's frames-per-call budget shrunk, and the number of frames already issued
    comes back so that the halving loop carries on from there."""
    import types
    import torch
    from cama_amd.dataset import ClipManager
    calls, shrunk = [], []

    def launch(desc, w2c_ptr, F, src_ptr, out_ptr, keep):
        if len(calls) == 2:
            raise torch.OutOfMemoryError("libcama_hip error -3: hipMalloc of pipeline scratch")
        calls.append(F)
    eng = types.SimpleNamespace(render_clip_launch=launch, shrink_frames_per_call=lambda: shrunk.append(1))
    cm = ClipManager.__new__(ClipManager)
    memo = {"desc": object(), "launches": [(0, 16, 0, 0, ()), (64 * 16, 16, 0, 0, ()), (64 * 32, 8, 0, 0, ())]}
    cm._launch_memo = {"cama": memo}
    assert cm._replay_launches(eng, "cama", memo) == 32 and calls == [16, 16] and shrunk == [1] and "cama" not in cm._launch_memo
    calls.clear()
    eng.render_clip_launch = lambda desc, w, F, s, o, k: calls.append(F)
    cm._launch_memo = {"cama": memo}
    assert cm._replay_launches(eng, "cama", memo) is None and calls == [16, 16, 8] and "cama" in cm._launch_memo


def test_rank_cpu_sets_are_disjoint_and_follow_the_gpus_numa_nodes():
    """one process per GPU, each pinned to cores of its GPU's NUMA node."""
    from cama_amd import shard
    two_nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    sets = shard.rank_cpu_sets(range(256), 8, [0, 0, 0, 0, 1, 1, 1, 1], two_nodes)
    assert all(len(s) == 32 for s in sets)
    assert len(set().union(*map(set, sets))) == 256              # disjoint and complete
    assert all(set(s) <= set(two_nodes[0]) for s in sets[:4]) and all(set(s) <= set(two_nodes[1]) for s in sets[4:])
    # restricted affinity (a cgroup / taskset): only allowed cores are handed out
    sets = shard.rank_cpu_sets(range(0, 64), 2, [0, 1], two_nodes)
    assert sets == [list(range(0, 32)), list(range(32, 64))]    # node 1 has no allowed core: even split of what is allowed
    # no NUMA picture: even split; fewer cores than ranks: leave the affinity alone
    assert shard.rank_cpu_sets(range(16), 4) == [list(range(k * 4, k * 4 + 4)) for k in range(4)]
    assert shard.rank_cpu_sets([3], 2) == [[], []]
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
