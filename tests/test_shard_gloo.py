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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        costs = [shard.scene_cost(40, 10000 + 100 * i, 160, 90) for i in range(5)]
        mine = shard.assign_scenes(costs, world)[rank]
        # each rank "renders" its scenes: here a deterministic byte pattern per scene stands in for the mosaic
        frames, lo, hi = 0, 0, 0
        for s in mine:
            m = (torch.arange(64 * 48 * 3, dtype=torch.int64) * (s + 3) % 251).to(torch.uint8)
            a, b = shard.overlay_hash(m)
            lo ^= a
            hi ^= b
            frames += 40
        rec = [frames, 0.5 + 0.25 * rank, 1.0, 1.0, 10000, sum(costs[s] for s in mine), float(lo % 2 ** 52), float(hi % 2 ** 52)]
        allrec = shard.gather_records(rec)
        q.put((rank, mine, shard.reduce_metrics(allrec)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_and_reduce():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, m0), (r1, s1, m1) = got
    assert sorted(s0 + s1) == [0, 1, 2, 3, 4] and not set(s0) & set(s1)
    assert m0 == m1                                         # every rank sees the same aggregate
    assert m0["frames"] == 200 and m0["seconds"] == 0.75 and m0["world"] == 2
    assert m0["frames_per_s"] == pytest.approx(200 / 0.75)


def test_overlay_hash_detects_a_single_byte():
    import torch
    a = torch.arange(1000, dtype=torch.int64).to(torch.uint8)
    b = a.clone()
    b[777] ^= 1
    assert shard.overlay_hash(a) != shard.overlay_hash(b)
    assert shard.overlay_hash(a) == shard.overlay_hash(a.numpy())
