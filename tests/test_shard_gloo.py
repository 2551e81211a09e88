"""Multi-process path on CPU: scene sharding + the one metric all_gather, gloo backend, world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest

from cama_amd import shard


def test_assign_scenes_lpt_and_ranges():
    assert shard.assign_scenes([1] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]          # equal costs: round-robin
    parts = shard.assign_scenes([1.0] * 73, 8)
    assert sorted(i for p in parts for i in p) == list(range(73))
    assert max(len(p) for p in parts) == 10 and min(len(p) for p in parts) == 9          # 73 scenes / 8 GPUs
    parts = shard.assign_scenes([10, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], 2)
    assert parts[0] == [0] and len(parts[1]) == 10                                       # one heavy scene alone
    assert shard.frame_ranges(1000, 8)[0] == (0, 125) and shard.frame_ranges(1000, 8)[-1] == (875, 1000)
    assert shard.frame_ranges(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert shard.scene_cost(40, 10000, 1600, 900) == 40 * (13 * 10000 + 36 * 1600 * 900)


def _fake_mosaic(scene):
    import torch
    return (torch.arange(64 * 48 * 3, dtype=torch.int64) * (scene + 3) % 251).to(torch.uint8)


def _worker(rank, world, port, q, corrupt_scene):
    """What bench.py does after its timed region, on CPU tensors over gloo: hash every scene this rank rendered, pack
    metrics + per-scene hashes into ONE int64 report, all_gather it, unpack, reduce and verify against golden hashes."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_scenes = 5
        costs = [shard.scene_cost(40, 10000 + 100 * i, 160, 90) for i in range(n_scenes)]
        mine = shard.assign_scenes(costs, world)[rank]
        # each rank "renders" its scenes: a deterministic byte pattern per scene stands in for the mosaic
        hashes = []
        for s in mine:
            m = _fake_mosaic(s)
            if s == corrupt_scene:
                m[100] ^= 1                                     # one wrong byte in one scene of one rank
            hashes.append((s,) + shard.overlay_hash(m))
        metrics = [40.0 * len(mine), 0.5 + 0.25 * rank, 1.0, 1.0, 10000.0, sum(costs[s] for s in mine), 40.0, 0.0]
        slots = -(-n_scenes // world) + 1
        allrep = shard.gather_reports(shard.pack_report(metrics, hashes, slots))
        m, found, owner = shard.unpack_reports(allrep, len(metrics))
        golden = {s: shard.overlay_hash_np(_fake_mosaic(s).numpy()) for s in range(n_scenes)}
        check = shard.verify_hashes(found, golden, expect_units=range(n_scenes))
        q.put((rank, mine, shard.reduce_metrics(m), check, owner, dist.get_world_size()))
    finally:
        dist.destroy_process_group()


def _run_two_ranks(corrupt_scene):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, corrupt_scene)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_rank_gloo_gather_reduce_and_hash_check():
    (r0, s0, m0, c0, o0, w0), (r1, s1, m1, c1, o1, w1) = _run_two_ranks(corrupt_scene=-1)
    assert sorted(s0 + s1) == [0, 1, 2, 3, 4] and not set(s0) & set(s1)
    assert m0 == m1 and c0 == c1 and w0 == w1 == 2         # every rank sees the same aggregate and verdict
    assert m0["frames"] == 200 and m0["seconds"] == 0.75 and m0["world"] == 2
    assert m0["frames_per_s"] == pytest.approx(200 / 0.75)
    assert c0 == {"verified": 5, "unverified": [], "mismatched": [], "missing": []}
    assert {s: o0[s] for s in s0} == {s: 0 for s in s0} and {s: o0[s] for s in s1} == {s: 1 for s in s1}


def test_two_rank_gloo_wrong_scene_is_caught():
    """A single wrong byte in one scene on one rank: the per-scene comparison with the golden hashes names it."""
    (_, _, _, c0, _, _), _ = _run_two_ranks(corrupt_scene=3)
    assert c0["mismatched"] == [3] and c0["verified"] == 4


def test_report_roundtrip_and_verdicts():
    big = (1 << 64) - 1
    r0 = shard.pack_report([1.5, -2.25], [(3, big, 5), (7, 1, 1 << 63)], 4)
    r1 = shard.pack_report([0.5, 9.0], [(4, 6, 7)], 4)
    m, found, owner = shard.unpack_reports(np.stack([r0, r1]), 2)
    assert m.tolist() == [[1.5, -2.25], [0.5, 9.0]]
    assert found == {3: (big, 5), 7: (1, 1 << 63), 4: (6, 7)} and owner == {3: 0, 7: 0, 4: 1}
    v = shard.verify_hashes(found, {3: (big, 5), 4: (6, 8)}, expect_units=[3, 4, 7, 9])
    assert v == {"verified": 1, "unverified": [7], "mismatched": [4], "missing": [9]}
    with pytest.raises(RuntimeError):                       # the same scene reported by two ranks
        shard.unpack_reports(np.stack([r0, shard.pack_report([0.0, 0.0], [(3, 1, 1)], 4)]), 2)


def test_committed_golden_hashes_cover_the_bench_workloads(repo_root):
    import bench
    path = os.path.join(repo_root, "tests", "golden", "scene_hashes.json")
    a = bench.parse_args([])
    sweep = shard.load_golden_hashes(path, bench.workload_key(a.frames, a.verts, a.width, a.height, "lanes"))
    assert sorted(sweep) == list(range(bench.SWEEP_SCENES))                  # configs[2]: all 73 scenes
    assert len(set(sweep.values())) == bench.SWEEP_SCENES                     # all distinct
    stress = shard.load_golden_hashes(path, bench.workload_key(bench.STRESS["frames"], bench.STRESS["verts"], a.width,
                                                               a.height, "random", unit="frame"))
    assert sorted(stress) == bench.stress_sample_frames(bench.STRESS["frames"])


def test_frame_pattern_is_device_independent():
    import torch
    from cama_amd.synth import frame_pattern, frame_pattern_np
    a = frame_pattern_np(7, (3, 5, 8, 3))
    assert np.array_equal(a, frame_pattern(7, (3, 5, 8, 3), "cpu", chunk_bytes=64).numpy())
    assert np.array_equal(frame_pattern_np(7, (2, 5, 8, 3), first=5 * 8 * 3), a[1:])
    assert np.array_equal(frame_pattern(7, (2, 5, 8, 3), "cpu", first=5 * 8 * 3).numpy(), a[1:])
    assert not np.array_equal(frame_pattern_np(8, (3, 5, 8, 3)), a)


def test_overlay_hash_detects_a_single_byte():
    import torch
    a = torch.arange(1000, dtype=torch.int64).to(torch.uint8)
    b = a.clone()
    b[777] ^= 1
    assert shard.overlay_hash(a) != shard.overlay_hash(b)
    assert shard.overlay_hash(a) == shard.overlay_hash(a.numpy()) == shard.overlay_hash_np(a.numpy())
    # position-weighted: swapping two rows changes the hash although the plain sum stays the same
    c = a.reshape(125, 8).clone()
    c[[3, 4]] = c[[4, 3]]
    assert shard.overlay_hash(c)[0] == shard.overlay_hash(a)[0] and shard.overlay_hash(c)[1] != shard.overlay_hash(a)[1]


# ---------------------------------------------------------------------------------------------------------------
# site affinity (BASELINE configs[3]: scenes of one site share the site's static vertex buffer)
# ---------------------------------------------------------------------------------------------------------------
def test_assign_scenes_keeps_a_site_on_as_few_ranks_as_balance_allows():
    site_of = [k // 6 for k in range(24)]                                   # 4 sites x 6 scenes
    two = shard.assign_scenes([1.0] * 24, 2, site_of=site_of, site_cost=0.3)
    assert shard.sites_per_rank(two, site_of) == [[0, 2], [1, 3]]           # whole sites, two per rank
    eight = shard.assign_scenes([1.0] * 24, 8, site_of=site_of, site_cost=0.3)
    assert [len(p) for p in eight] == [3] * 8                               # more ranks than sites: each site on 2 ranks
    assert all(len(s) == 1 for s in shard.sites_per_rank(eight, site_of))
    # no site id: the plain longest-processing-time-first assignment, unchanged
    assert shard.assign_scenes([1.0] * 73, 8, site_of=None) == shard.assign_scenes([1.0] * 73, 8)
    # every scene exactly once, balance within 15 % of the mean, for uneven costs and awkward site counts
    rng = np.random.default_rng(0)
    for n, world, n_sites in ((73, 8, 10), (73, 8, 3), (40, 4, 7), (9, 8, 2)):
        costs = rng.uniform(0.5, 2.0, n)
        site_of = [k % n_sites for k in range(n)]
        parts = shard.assign_scenes(costs, world, site_of=site_of, site_cost=0.2)
        assert sorted(i for p in parts for i in p) == list(range(n))
        loads = [costs[p].sum() + 0.2 * len(s) for p, s in zip(parts, shard.sites_per_rank(parts, site_of))]
        if n >= 4 * world:
            assert max(loads) <= 1.15 * np.mean(loads), (n, world, n_sites, loads)
        # affinity: far fewer (rank, site) pairs than the site-blind assignment produces
        blind = shard.assign_scenes(costs, world)
        pairs = sum(len(s) for s in shard.sites_per_rank(parts, site_of))
        assert pairs <= sum(len(s) for s in shard.sites_per_rank(blind, site_of))


def test_assign_scenes_never_parks_the_job_on_one_rank():
    # a dominant site load must not collapse a site (here: the whole job) onto one rank while seven idle
    assert shard.assign_scenes([1.0] * 8, 8, site_of=[0] * 8, site_cost=100.0) == [[k] for k in range(8)]
    # all-zero costs: round-robin, with and without sites
    assert shard.assign_scenes([0.0] * 8, 4) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert shard.assign_scenes([0.0] * 8, 4, site_of=[0] * 8, site_cost=0.0) == [[0, 4], [1, 5], [2, 6], [3, 7]]
    # the affinity never loses to the plain longest-first placement once every rank pays for the sites it touches
    rng = np.random.default_rng(1)
    for _ in range(500):
        n, world, n_sites = int(rng.integers(1, 30)), int(rng.integers(1, 9)), int(rng.integers(1, 6))
        costs, site_of, sc = rng.uniform(0, 10, n), rng.integers(0, n_sites, n).tolist(), float(rng.uniform(0, 50))
        span = lambda parts: max(sum(costs[i] for i in p) + sc * len({site_of[i] for i in p}) for p in parts)
        got = shard.assign_scenes(costs, world, site_of=site_of, site_cost=sc)
        assert sorted(i for p in got for i in p) == list(range(n))
        assert span(got) <= span(shard.assign_scenes(costs, world)) * (1 + 1e-12)


def _site_worker(rank, world, port, q):
    """Two ranks, 2 sites x 4 scenes: every rank builds the site maps of ITS scenes only (a set, as the content-keyed
    device cache would hold them), "renders" its scenes against them, and the usual single all_gather carries the scene
    hashes plus the number of site maps the rank had to load."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_scenes, n_sites = 8, 2
        site_of = [k % n_sites for k in range(n_scenes)]
        costs = [shard.scene_cost(6, 60000, 320, 180)] * n_scenes
        mine = shard.assign_scenes(costs, world, site_of=site_of, site_cost=40.0 * 60000)[rank]
        loaded = {}
        hashes = []
        for s in mine:
            site = site_of[s]
            if site not in loaded:                              # one "upload" per site per rank
                loaded[site] = (torch.arange(4096, dtype=torch.int64) * (site + 11) % 253).to(torch.uint8)
            m = _fake_mosaic(s) ^ loaded[site][:_fake_mosaic(s).numel() % 4096 + 1].sum().to(torch.uint8)
            hashes.append((s,) + shard.overlay_hash(m))
        metrics = [6.0 * len(mine), 1.0, float(len(loaded)), 0.0]
        allrep = shard.gather_reports(shard.pack_report(metrics, hashes, n_scenes))
        m, found, owner = shard.unpack_reports(allrep, len(metrics))
        q.put((rank, mine, sorted(loaded), m[:, 2].tolist(), sorted(found), owner))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_site_affinity():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_site_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, l0, up0, f0, o0), (r1, s1, l1, up1, f1, o1) = got
    assert sorted(s0 + s1) == list(range(8)) and f0 == f1 == list(range(8))
    assert len(l0) == 1 and len(l1) == 1 and l0 != l1           # each rank loaded exactly one site, a different one
    assert up0 == up1 == [1.0, 1.0]                             # ... and the gathered report says so on every rank
    assert {k % 2 for k in s0} == set(l0) and {k % 2 for k in s1} == set(l1)
