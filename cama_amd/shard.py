"""Multi-GPU sharding of the reprojection workload.

Scenes (clips) are independent -- main.py's `for scene_name in configs["scene_names"]` loop carries no state
across iterations (main.py:32) -- so the unit of distribution is the scene, one process per GPU, and there is NO
data-path collective.  A single long scene can additionally be cut into contiguous frame ranges (frames are
independent given the pose track).  The only communication is one all_gather of a fixed-size metric record per
run (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
import numpy as np

RECORD_FIELDS = ("frames", "seconds", "overlay_ms", "overlay_launches", "verts", "bytes", "frames_per_launch", "aux")


def assign_scenes(costs, world, site_of=None, site_cost=0.0):
    """Longest-processing-time-first assignment of scenes to `world` ranks.

    costs: per-scene cost estimate, e.g. F * (13 N + 36 W H) bytes.  Returns a list of `world` lists of scene
    indices (each sorted).  Deterministic; ties broken by scene index.  Round-robin falls out for equal costs.

    site_of (optional, one site id per scene): scenes of one site share the site's static vertex buffer (CAMA v2's
    site-aggregated labels) and every rank that renders one of them has to load, sort and index that
    buffer (`site_cost`, in the units of `costs`, per (rank, site)).  Scenes are therefore placed site by site: a
    site goes to the least loaded rank, which takes as many of its scenes as fit under the fair share of the job; only
    what does not fit moves on to another rank (and pays the site's load again)."""
    costs = np.asarray(costs, np.float64)
    if site_of is None:
        order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
        load = [0.0] * world
        out = [[] for _ in range(world)]
        for i in order:
            r = min(range(world), key=lambda k: (load[k], len(out[k]), k))    # (zero costs: fewest scenes first)
            out[r].append(i)
            load[r] += float(costs[i])
        return [sorted(s) for s in out]
    site_of = list(site_of)
    assert len(site_of) == len(costs)
    groups = {}
    for i, sid in enumerate(site_of):
        groups.setdefault(sid, []).append(i)

    def makespan(assignment):
        return max((sum(float(costs[i]) for i in sc) + site_cost * len({site_of[i] for i in sc})) for sc in assignment) \
            if assignment else 0.0

    plain = assign_scenes(costs, world)                       # no affinity: every rank may load every site
    # fair share of a rank, counting one load per site and at least one load per rank
    fair = (float(costs.sum()) + site_cost * max(len(groups), world)) / world
    if fair <= 0.0:                                           # nothing to balance on: round-robin
        return plain
    site_total = {sid: float(costs[m].sum()) for sid, m in groups.items()}
    load = [0.0] * world
    out = [[] for _ in range(world)]
    # sites longest first; each goes to the least loaded rank, which takes as many of its scenes as fit under the fair
    # share (always at least one); what does not fit moves on to the next least loaded rank, paying the load again.  With
    # fewer sites than ranks a rank takes at most its even share of a site's scenes -- a dominant site_cost must not park
    # a whole site (in the limit the whole job) on one rank while the others idle.
    for sid in sorted(groups, key=lambda g: (-site_total[g], groups[g][0])):
        todo = sorted(groups[sid], key=lambda i: (-costs[i], i))
        cap = len(todo) if len(groups) >= world else max(1, -(-len(todo) * len(groups) // world))
        while todo:
            r = min(range(world), key=lambda k: (load[k], k))
            load[r] += site_cost
            took = 0
            rest = float(costs[todo].sum())
            while todo and took < cap and (took == 0 or load[r] + float(costs[todo[0]]) <= 1.02 * fair or
                                           load[r] + rest <= 1.10 * fair):
                i = todo.pop(0)
                out[r].append(i)
                load[r] += float(costs[i])
                rest -= float(costs[i])
                took += 1
    out = [sorted(s) for s in out]
    # the affinity is a heuristic: never accept it when the plain longest-first placement (each rank paying for every
    # site it touches) finishes earlier
    return out if makespan(out) <= makespan(plain) else plain


def sites_per_rank(assignment, site_of):
    """[sorted site ids each rank has to load] for an assign_scenes() result."""
    return [sorted({site_of[i] for i in scenes}) for scenes in assignment]


def frame_ranges(n_frames, world):
    """Contiguous, balanced [lo, hi) frame ranges of one long scene, one per rank (empty ranges allowed)."""
    base, extra = divmod(n_frames, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def scene_cost(n_frames, n_verts, W, H, cams=6):
    """Algorithmic HBM bytes of a scene: F * (13 N + 2*cams*W*H*3)  (SURVEY.md 8d)."""
    return float(n_frames) * (13.0 * n_verts + 2.0 * cams * W * H * 3.0)


_HASH_MULT = 0x9E3779B97F4A7C15                 # odd 64-bit constant (golden ratio)


def overlay_hash_np(arr, chunk_words=1 << 22):
    """numpy twin of overlay_hash (what the oracle side hashes with): (lo, hi) uint64 as Python ints."""
    b = np.ascontiguousarray(arr).reshape(-1).view(np.uint8)
    n8 = (b.size // 8) * 8
    w = b[:n8].view(np.uint64)
    tail = int(b[n8:].astype(np.uint64).sum())
    lo = hi = 0
    with np.errstate(over="ignore"):
        for a in range(0, w.size, chunk_words):
            c = w[a:a + chunk_words]
            idx = np.arange(a, a + c.size, dtype=np.uint64)
            lo += int(c.sum(dtype=np.uint64))
            hi += int((c * ((idx * np.uint64(2) + np.uint64(1)) * np.uint64(_HASH_MULT))).sum(dtype=np.uint64))
    return (lo + tail) & 0xFFFFFFFFFFFFFFFF, (hi + 31 * tail) & 0xFFFFFFFFFFFFFFFF


def overlay_hash(mosaic, chunk_words=1 << 24):
    """128-bit checksum of a uint8 device/host tensor: lo = sum of its little-endian 64-bit words, hi = sum of
    word_i * (2 i + 1) * K (both mod 2^64; K odd).  hi is position-weighted: moving, swapping or dropping rows,
    frames or camera cells changes it, which a plain sum would not notice.  Integer arithmetic only, so the value is
    the same on every device and equals overlay_hash_np on the same bytes."""
    import torch
    t = mosaic if isinstance(mosaic, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mosaic))
    flat = t.reshape(-1)
    n8 = (flat.numel() // 8) * 8
    words = flat[:n8].view(torch.int64) if n8 else flat[:0].to(torch.int64)
    mult = _HASH_MULT - (1 << 64)                   # the same 64 bits as a signed value; int64 products wrap mod 2^64
    lo = hi = 0
    for a in range(0, words.numel(), chunk_words):
        w = words[a:a + chunk_words]
        idx = torch.arange(a, a + w.numel(), dtype=torch.int64, device=w.device)
        lo += int(w.sum().item())
        hi += int((w * ((idx * 2 + 1) * mult)).sum().item())
    tail = int(flat[n8:].to(torch.int64).sum().item())
    return (lo + tail) & 0xFFFFFFFFFFFFFFFF, (hi + 31 * tail) & 0xFFFFFFFFFFFFFFFF


def gather_records(record, device=None):
    """all_gather one float64 record per rank -> (world, len(record)) numpy array on every rank.
    Works with the nccl (= RCCL) backend on GPU tensors and with gloo on CPU tensors; world 1 needs no group."""
    import torch
    import torch.distributed as dist
    rec = torch.as_tensor(record, dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return rec[None].cpu().numpy()
    world = dist.get_world_size()
    out = torch.empty(world * rec.numel(), dtype=torch.float64, device=rec.device)    # flat: accepted by nccl and gloo
    dist.all_gather_into_tensor(out, rec)
    return out.reshape(world, rec.numel()).cpu().numpy()


# ---------------------------------------------------------------------------------------------------------------
# the bench's one collective: per-rank report = float64 metrics + (unit id, hash lo, hash hi) rows, as int64 words
# ---------------------------------------------------------------------------------------------------------------
def _to_i64(u):
    u = int(u) & 0xFFFFFFFFFFFFFFFF
    return u - (1 << 64) if u >= (1 << 63) else u


def pack_report(metrics, hashes, slots):
    """One rank's report as a flat int64 array: the float64 `metrics` bit-cast to int64, then `slots` rows of
    (unit id, hash lo, hash hi); unused rows carry id -1.  `hashes` = [(unit id, lo, hi)] of the units (scenes, or
    sampled frames of a frame-sharded scene) this rank rendered."""
    assert len(hashes) <= slots, (len(hashes), slots)
    rows = np.full((slots, 3), -1, np.int64)
    for k, (uid, lo, hi) in enumerate(hashes):
        rows[k] = (int(uid), _to_i64(lo), _to_i64(hi))
    return np.concatenate([np.asarray(metrics, np.float64).view(np.int64), rows.reshape(-1)])


def gather_reports(report, device=None):
    """all_gather of one pack_report() array per rank -> (world, len) int64 numpy array on every rank.  This is the
    job's only collective (nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests; world 1 needs no group)."""
    import torch
    import torch.distributed as dist
    rec = torch.as_tensor(np.ascontiguousarray(report), dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return rec[None].cpu().numpy()
    world = dist.get_world_size()
    out = torch.empty(world * rec.numel(), dtype=torch.int64, device=rec.device)
    dist.all_gather_into_tensor(out, rec)
    return out.reshape(world, rec.numel()).cpu().numpy()


def unpack_reports(reports, n_metrics):
    """(world, L) int64 -> (metrics float64 (world, n_metrics), {unit id: (lo, hi)}, {unit id: rank})."""
    r = np.ascontiguousarray(np.asarray(reports, np.int64))
    metrics = np.ascontiguousarray(r[:, :n_metrics]).view(np.float64)
    found, owner = {}, {}
    for rank in range(r.shape[0]):
        for uid, lo, hi in r[rank, n_metrics:].reshape(-1, 3):
            if uid >= 0:
                if int(uid) in found:
                    raise RuntimeError(f"unit {int(uid)} was rendered by ranks {owner[int(uid)]} and {rank}")
                found[int(uid)] = (int(lo) & 0xFFFFFFFFFFFFFFFF, int(hi) & 0xFFFFFFFFFFFFFFFF)
                owner[int(uid)] = rank
    return metrics, found, owner


def verify_hashes(found, golden, expect_units=None):
    """Compare {unit: (lo, hi)} with the committed golden {unit: (lo, hi)} (oracle-rendered, tests/golden/*.json).
    Returns {"verified": n, "unverified": [units without a golden entry], "mismatched": [...], "missing": [units
    that should have been rendered and were not]}."""
    golden = {int(k): (int(v[0]), int(v[1])) for k, v in golden.items()}
    out = {"verified": 0, "unverified": [], "mismatched": [], "missing": []}
    for uid in sorted(found):
        if uid not in golden:
            out["unverified"].append(uid)
        elif golden[uid] == tuple(found[uid]):
            out["verified"] += 1
        else:
            out["mismatched"].append(uid)
    if expect_units is not None:
        out["missing"] = sorted(set(int(u) for u in expect_units) - set(found))
    return out


def load_golden_hashes(path, key):
    """{unit: (lo, hi)} of the golden file's entry `key` (the workload description), or {} when absent."""
    import json
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return {}
    return {int(k): (int(v[0], 16), int(v[1], 16)) for k, v in rec.get(key, {}).items()}


def reduce_metrics(records):
    """records (world, 8) as RECORD_FIELDS -> whole-job aggregate: frames summed, seconds = max over ranks."""
    r = np.asarray(records, np.float64)
    frames, seconds = float(r[:, 0].sum()), float(r[:, 1].max())
    return {"frames": frames, "seconds": seconds, "frames_per_s": frames / seconds if seconds > 0 else 0.0,
            "bytes": float(r[:, 5].sum()), "world": int(r.shape[0])}


# ---------------------------------------------------------------------------------------------------------------
# host placement of the ranks: one process per GPU, each on cores of its GPU's NUMA node
# ---------------------------------------------------------------------------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> sorted list of ints."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return sorted(set(out))


def rank_cpu_sets(allowed, world, node_of_rank=None, cpus_of_node=None):
    """Disjoint CPU sets for the `world` ranks of one node.

    allowed        the cores this job may use (the launcher's scheduler affinity)
    node_of_rank   NUMA node of each rank's GPU (None / -1: unknown)
    cpus_of_node   {node: cores of that node}
    A rank gets a contiguous share of (its GPU's node's cores that are allowed), split among the ranks whose GPUs sit on
    the same node; when any rank's node is unknown, or a node has fewer allowed cores than ranks, every rank gets an even
    contiguous share of the allowed cores instead.  Returns a list of `world` sorted lists; empty when there is nothing sensible to pin to (fewer
    allowed cores than ranks): the caller then leaves the affinity alone.  Pure function (tests)."""
    allowed = sorted(set(int(c) for c in allowed))
    if world < 1 or len(allowed) < world:
        return [[] for _ in range(max(world, 0))]
    even = [allowed[k * (len(allowed) // world):(k + 1) * (len(allowed) // world)] for k in range(world)]
    node_of_rank = list(node_of_rank) if node_of_rank is not None else [None] * world
    cpus_of_node = cpus_of_node or {}
    if any(n is None or n < 0 or n not in cpus_of_node for n in node_of_rank):
        return even                                     # no (complete) NUMA picture: an even split of what is allowed
    by_node = {}
    for r, n in enumerate(node_of_rank):
        by_node.setdefault(n, []).append(r)
    out = [None] * world
    ok = set(allowed)
    for n, ranks in sorted(by_node.items()):
        cores = [c for c in sorted(cpus_of_node[n]) if c in ok]
        if len(cores) < len(ranks):
            return even                                 # (a node whose cores the job may not use)
        per = len(cores) // len(ranks)
        for k, r in enumerate(ranks):
            out[r] = cores[k * per:(k + 1) * per]
    return out


def gpu_numa_node(pci_bus_id):
    """NUMA node of the GPU at PCI address 'dddd:bb:dd.f' (sysfs), or -1."""
    try:
        return int(open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node").read().strip())
    except (OSError, ValueError):
        return -1


def node_cpus():
    """{NUMA node: cores} from sysfs ({} when the host does not expose it)."""
    import glob
    import os
    out = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try:
            out[int(os.path.basename(d)[4:])] = parse_cpulist(open(os.path.join(d, "cpulist")).read())
        except (OSError, ValueError):
            pass
    return out


def bind_rank(local_rank, local_world, pci_bus_ids=None, allowed=None):
    """Pin the calling process (one rank of `local_world` on this node) to its share of the cores next to its GPU.
    pci_bus_ids: PCI address of every local rank's GPU (rank order), or None (no NUMA information: an even split of the
    allowed cores).  Returns {"cpus": [...], "numa_node": n, "bound": bool}; never raises (placement is speed only)."""
    import os
    info = {"cpus": [], "numa_node": -1, "bound": False}
    try:
        allowed = sorted(os.sched_getaffinity(0)) if allowed is None else sorted(allowed)
        nodes = [gpu_numa_node(b) if b else -1 for b in (pci_bus_ids or [None] * local_world)]
        sets = rank_cpu_sets(allowed, local_world, nodes, node_cpus())
        mine = sets[local_rank] if local_rank < len(sets) else []
        info["numa_node"] = nodes[local_rank] if local_rank < len(nodes) else -1
        if mine:
            os.sched_setaffinity(0, mine)
            info["cpus"], info["bound"] = mine, True
        else:
            info["cpus"] = allowed
    except (OSError, AttributeError, ValueError):
        pass
    return info
