"""Multi-GPU sharding of the reprojection workload.

Scenes (clips) are independent -- main.py's `for scene_name in configs["scene_names"]` loop carries no state
across iterations (main.py:32) -- so the unit of distribution is the scene, one process per GPU, and there is NO
data-path collective.  A single long scene can additionally be cut into contiguous frame ranges (frames are
independent given the pose track).  The only communication is one all_gather of a fixed-size metric record per
run (RCCL over xGMI on the GPU box, gloo in the CPU tests).
"""
import numpy as np

RECORD_FIELDS = ("frames", "seconds", "overlay_ms", "overlay_launches", "verts", "bytes", "hash_lo", "hash_hi")


def assign_scenes(costs, world):
    """Longest-processing-time-first assignment of scenes to `world` ranks.

    costs: per-scene cost estimate, e.g. F * (13 N + 36 W H) bytes.  Returns a list of `world` lists of scene
    indices (each sorted).  Deterministic; ties broken by scene index.  Round-robin falls out for equal costs."""
    costs = np.asarray(costs, np.float64)
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(s) for s in out]


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


def overlay_hash(mosaic):
    """Order-independent 128-bit checksum of a uint8 device/host tensor (two 64-bit sums of two views)."""
    import torch
    t = mosaic if isinstance(mosaic, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(mosaic))
    flat = t.reshape(-1)
    n8 = (flat.numel() // 8) * 8
    words = flat[:n8].view(torch.int64)
    lo = int(words.sum().item()) & 0xFFFFFFFFFFFFFFFF
    hi = int((words >> 7).sum().item() + flat[n8:].to(torch.int64).sum().item()) & 0xFFFFFFFFFFFFFFFF
    return lo, hi


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


def reduce_metrics(records):
    """records (world, 8) as RECORD_FIELDS -> whole-job aggregate: frames summed, seconds = max over ranks."""
    r = np.asarray(records, np.float64)
    frames, seconds = float(r[:, 0].sum()), float(r[:, 1].max())
    return {"frames": frames, "seconds": seconds, "frames_per_s": frames / seconds if seconds > 0 else 0.0,
            "bytes": float(r[:, 5].sum()), "world": int(r.shape[0]),
            "hash": [[int(a), int(b)] for a, b in zip(r[:, 6], r[:, 7])]}
