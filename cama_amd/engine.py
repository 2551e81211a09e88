"""Device-side engine: owns the HBM-resident buffers of one clip and drives libcama_hip.so.

PyTorch is used for device memory, streams and (in bench.py) torch.distributed only; every
computation below is a call through the C ABI (include/cama_hip.h).  There is no CPU fallback:
constructing an Engine without a GPU or without the built library raises.

Reference semantics implemented by the kernels: cama/dataset.py:99-117, cama/reproject.py:108-131,
187-205,246-257, cama/tools.py:22-25 (paths under /root/reference).
"""
import numpy as np
import os

from . import _lib

CROP_BOX = (-50.0, 50.0, -100.0, 100.0, -200.0, 200.0)        # cama/reproject.py:28-34
PALETTE_BGR = ((211, 211, 211), (0, 215, 255))                # grey lane_marking, gold everything else
RADIUS = 2                                                    # cama/reproject.py:256
# maps with at least this many vertices hand their block index (per-block AABBs) to the render: below, the
# one-thread-per-block pre-pass costs a launch for nothing.  CAMA_TEST_HOOKS=bounds_min_verts=1 forces it (tests).
_HOOKS = _lib.test_hooks()          # CAMA_TEST_HOOKS (the fuzz tests' child processes)
BOUNDS_MIN_VERTS = int(_HOOKS.get("bounds_min_verts", 65536))   # maps from this size get the per-block spatial index
PIPELINE_DEPTH = 2                  # scratch slots of a cama_pipeline (cama_hip.hip; three measured the same everywhere, round 5)
USE_BOUNDS = not _HOOKS.get("no_bounds")    # test hook: no spatial index
USE_RAW35 = True                    # test hook: False = the general raw-frame kernels instead of the 3:5 one
PLAN_SITE_MAPS = True               # test hook: False = worst-case scratch for site-sized maps
SITE_FRAMES_PER_LAUNCH = 128        # frames per launch of site-sized (planned) maps
MAX_SCENES_PER_LAUNCH = 1024                                  # include/cama_hip.h CAMA_MAX_SCENES_PER_LAUNCH


def _torch():
    import torch
    return torch


def _content_hasher():
    """128-bit content hash for Engine.shared_map's cache key: xxh3 when the xxhash package is there (10x faster on a
    4e6-vertex site map), else hashlib's blake2b -- neither the reference's requirements.txt nor ours demands xxhash.  The
    key only has to be deterministic inside one process."""
    try:
        import xxhash
        return xxhash.xxh3_128()
    except ImportError:
        import hashlib
        return hashlib.blake2b(digest_size=16)


class host_keepalive:
    """Holds a host array alive inside a pipeline keep tuple (the library reads the scene table's host copy only during
    the call, but the cache may evict it while launches are queued: keeping it costs nothing)."""

    def __init__(self, arr):
        self.arr = arr


class CameraRig:
    """Per-clip calibration on the device: [C,16] chassis->camera and [C,9] K (scaled to W x H)."""

    def __init__(self, names, chassis2camera, K, W, H, device):
        torch = _torch()
        self.names = list(names)
        self.C = len(self.names)
        self.W, self.H = int(W), int(H)
        c2c = np.ascontiguousarray(np.stack([np.asarray(m, np.float64).reshape(4, 4) for m in chassis2camera]))
        Ks = np.ascontiguousarray(np.stack([np.asarray(k, np.float64).reshape(3, 3) for k in K]))
        self.c2cam_host, self.K_host = c2c, Ks
        self.c2cam = torch.from_numpy(c2c.reshape(self.C, 16)).to(device)
        self.K = torch.from_numpy(Ks.reshape(self.C, 9)).to(device)


def morton_order(xyz):
    """Permutation that sorts points along a 2-D Morton (Z-order) curve over their x/y bounding box."""
    xy = np.asarray(xyz[:, :2], np.float64)
    lo = xy.min(axis=0)
    span = np.maximum(xy.max(axis=0) - lo, 1e-9)
    q = np.minimum(((xy - lo) / span * 65535.0).astype(np.uint64), np.uint64(65535))

    def spread(v):                                  # 16 bits -> every other bit of 32
        v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
        v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
        v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
        v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
        return v
    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)), kind="stable")


def lacks_spatial_order(xyz, threshold_m=2.0):
    """True when consecutive vertices are typically metres apart (polyline maps are centimetres apart)."""
    if xyz.shape[0] < 4096:
        return False
    step = np.abs(np.diff(np.asarray(xyz[:, :2], np.float64), axis=0)).max(axis=1)
    return float(np.median(step)) > threshold_m


class DeviceMap:
    """Static vertex buffer of one dataset pass: SoA x,y,z (float32 or float64) + colour id (uint8).

    Draw order = storage order (instance-major, point-minor), which is what the reference draws in.  Maps whose
    storage order has no spatial coherence (consecutive vertices metres apart) additionally get a Morton-sorted
    copy for the fused render: whole workgroups then fall inside / outside the crop box and hit few stamp bins,
    and the original index travels as the stamp key so "last writer wins" is unchanged."""

    def __init__(self, xyz, colour_id, device, spatial_sort="auto"):
        torch = _torch()
        xyz = np.asarray(xyz)
        if xyz.dtype not in (np.float32, np.float64):
            xyz = xyz.astype(np.float64)
        assert xyz.ndim == 2 and xyz.shape[1] == 3
        self.N = int(xyz.shape[0])
        self.is_f64 = int(xyz.dtype == np.float64)
        colour_id = np.ascontiguousarray(colour_id, dtype=np.uint8)
        assert colour_id.shape[0] == self.N
        self.soa = torch.from_numpy(np.ascontiguousarray(xyz.T)).to(device)          # [3,N]
        # bit 0 = palette index; bit 1 (optional, set by the caller: _StaticMap) = "joined to the previous vertex", read by
        # the segment extension only
        self.colour = torch.from_numpy(colour_id).to(device)
        self.has_links = bool((colour_id & 2).any())
        self.sorted_soa = self.sorted_key = None
        if spatial_sort is True or (spatial_sort == "auto" and lacks_spatial_order(xyz)):
            order = morton_order(xyz)
            key = ((order.astype(np.uint32) << np.uint32(1)) | (colour_id[order] & 1).astype(np.uint32))
            self.has_links = False                        # (a spatially sorted copy has no polyline neighbours)
            self.sorted_soa = torch.from_numpy(np.ascontiguousarray(xyz[order].T)).to(device)
            self.sorted_key = torch.from_numpy(np.ascontiguousarray(key)).to(device)
        self._index()

    # `bounds` / `extent_xy`: plain attributes once the index exists; a clip-sized map computes it on first access
    _index_pending = False

    @property
    def bounds(self):
        if self._index_pending:
            self._index(force=True)
        return self._bounds

    @bounds.setter
    def bounds(self, value):
        self._bounds = value
        self._index_pending = False

    @property
    def extent_xy(self):
        if self._index_pending:
            self._index(force=True)
        return self._extent_xy

    @extent_xy.setter
    def extent_xy(self, value):
        self._extent_xy = value

    def _index(self, force=False):
        """Spatial index for the fused render's crop step: per-block AABBs of the buffer it reads (cama_map_bounds,
        include/cama_hip.h) + the map's overall XY extent (host), which decides whether the index is worth using."""
        torch = _torch()
        self.bounds, self.extent_xy = None, (0.0, 0.0)
        if not self.N or not USE_BOUNDS:
            return
        if self.N < BOUNDS_MIN_VERTS and not force:
            # the render path asks for N >= BOUNDS_MIN_VERTS before it looks at the index: a clip-sized map (the usual 10^4
            # vertices) leaves the launch, the two reductions and -- what costs -- the blocking read-back of the extent (~1 ms
            # of a new clip's first frame, twice per scene: tools/cold_first_frame_profile.py) to whoever asks first (`bounds`)
            self._index_pending = True
            return
        L = _lib.lib()
        blk = L.cama_map_bounds_block()
        device = self.soa.device
        with torch.cuda.device(device):
            self.bounds = torch.empty(((self.N + blk - 1) // blk, 6), dtype=torch.float64, device=device)
            x, y, z = self.render_ptrs()[:3]
            _lib.check(L.cama_map_bounds(x, y, z, self.is_f64, self.N, self.bounds.data_ptr(),
                                         torch.cuda.current_stream(device).cuda_stream))
            lo = self.bounds[:, 0:4:2].min(dim=0).values
            hi = self.bounds[:, 1:4:2].max(dim=0).values
            ext = (hi - lo).cpu().numpy()
        self.extent_xy = tuple(float(e) if np.isfinite(e) and e > 0 else 0.0 for e in ext)

    def site_sized(self, crop):
        """True when the map's XY footprint is well beyond the crop box (site-aggregated maps: most vertex blocks are
        outside it on any frame) -- then the block cull pays; on clip-sized maps nearly every block survives and the
        plain grid launch is faster (measured: 29.8k vs 25.5k frames/s on 10^6 in-crop points)."""
        crop_area = float(crop[1] - crop[0]) * float(crop[3] - crop[2])
        return self.extent_xy[0] * self.extent_xy[1] > 2.0 * crop_area

    def ptrs(self):
        es = self.soa.element_size()
        base = self.soa.data_ptr()
        return base, base + self.N * es, base + 2 * self.N * es

    def render_ptrs(self, crop=None):
        """(x, y, z, colour, key, block bounds, bin flags) for the fused render: the sorted copy when there is one; the
        block bounds (per-camera / crop culling of whole vertex blocks) whenever the map has an index and is big enough
        for the one-thread-per-block pre-pass to pay; the work-list flag only for maps that are site-sized against `crop`."""
        bounds, flags = None, 0
        if crop is not None and self.N >= BOUNDS_MIN_VERTS and getattr(self, "bounds", None) is not None:
            bounds = self.bounds.data_ptr()
            if self.site_sized(crop):
                flags = _lib.BIN_WORKLIST
        if self.sorted_soa is None:
            return self.ptrs() + (self.colour.data_ptr(), None, bounds, flags)
        es = self.sorted_soa.element_size()
        base = self.sorted_soa.data_ptr()
        return (base, base + self.N * es, base + 2 * self.N * es, self.colour.data_ptr(), self.sorted_key.data_ptr(),
                bounds, flags)


class ChunkedMosaic:
    """F mosaic frames kept as separate device allocations of a few frames each (one per launch of a long clip), so that
    every launch's destination can be PLACED on its own (MosaicPool.take_many).  Indexing: an int gives a frame, a slice
    [lo:hi] a view when it stays inside one chunk (what ClipManager.render_clip asks for -- it cuts its launches at the
    chunk boundaries), anything else raises."""

    def __init__(self, chunks):
        self.chunks = list(chunks)
        assert self.chunks and all(c.shape[1:] == self.chunks[0].shape[1:] for c in self.chunks)
        self.bounds = [0]
        for c in self.chunks:
            self.bounds.append(self.bounds[-1] + int(c.shape[0]))

    @property
    def shape(self):
        return (self.bounds[-1],) + tuple(self.chunks[0].shape[1:])

    @property
    def device(self):
        return self.chunks[0].device

    def __len__(self):
        return self.bounds[-1]

    def spans(self):
        return [(self.bounds[k], self.bounds[k + 1], c) for k, c in enumerate(self.chunks)]

    def _chunk_of(self, f):
        import bisect
        if not 0 <= f < self.bounds[-1]:
            raise IndexError(f)
        return bisect.bisect_right(self.bounds, f) - 1

    def __getitem__(self, key):
        F = self.bounds[-1]
        if isinstance(key, slice):
            lo, hi, step = key.indices(F)
            if step != 1:
                raise IndexError("ChunkedMosaic: unit-stride slices only")
            if hi <= lo:
                return self.chunks[0][:0]
            k = self._chunk_of(lo)
            if hi > self.bounds[k + 1]:
                raise IndexError("ChunkedMosaic: [%d:%d] crosses the chunk boundary at %d" % (lo, hi, self.bounds[k + 1]))
            return self.chunks[k][lo - self.bounds[k]:hi - self.bounds[k]]
        f = int(key) + (F if int(key) < 0 else 0)
        k = self._chunk_of(f)
        return self.chunks[k][f - self.bounds[k]]

    def fill_(self, v):
        for c in self.chunks:
            c.fill_(v)
        return self

    def zero_(self):
        return self.fill_(0)


def _storage_users(t):
    """How many tensors / storages reference `t`'s memory besides the temporary this call creates."""
    import torch
    fn = getattr(torch._C, "_storage_Use_Count", None)
    if fn is None:
        return 1 << 30          # (a torch without the private counter: no base ever looks idle -- plain allocations, no recycling)
    return fn(t.untyped_storage()._cdata) - 1


class MosaicPool:
    """The engine's long-lived, PLACED mosaic buffers (placement belongs to the product).

    Every mosaic the product allocates on a caller's behalf -- ClipManager.render_clip(out=None), render_clips(outs=None),
    Engine.render_frames / render_frames_raw (out=None), the render-ahead batches behind render_vectors -- is a VIEW of a
    base buffer this pool owns.  A base is idle when nothing but the pool references its storage (the storage use count:
    views held by a caller, a RenderBatch or a pipelined launch's keep tuple all count), and an idle base of the right
    frame shape is handed out again instead of allocating: across batches, passes and ClipManagers a process renders into
    the same few buffers -- the reference's caller never manages buffers (cama/dataset.py:119-126) and need not here.

    Placement: the overlay's bandwidth depends on where source and destination sit physically relative to each other
    (profiles/r04_overlay_modes.txt section 5: 0.85 against 0.77 of 8 TB/s, the same buffers every time).  A NEW base of
    512 MiB .. 8 GiB whose source can be probed is therefore the fastest of CAMA_AUDITION (16) candidate allocations,
    timed with stamp-free overlay launches from that source (Engine._overlay_ms); runners-up within 3 % of the winner
    stay in the pool as idle bases (KEEP = 2 per audition: a pipelined caller ping-pongs between two
    buffers), the others go back to torch's allocator.  Paid once per shape per process.  CAMA_AUDITION=0: plain
    allocations, still pooled.  Speed only -- no option here can change a byte."""

    def __init__(self, engine):
        self.eng = engine
        self.bases = []                     # [{"t": tensor [F, ...frame shape], "frame": tuple, "ms": float|None, "stream": int, "lru": int}]
        self.clock = 0
        self.stats = {"takes": 0, "hits": 0, "allocations": 0, "auditions": 0, "trimmed": 0, "audition_seconds": 0.0,
                      "audition_peak_bytes": 0}
        self.flat_box = False               # an audition found every candidate alike: no further auditions in this process

    # ---- bookkeeping
    def _on_gpu(self):
        return _torch().device(self.eng.device).type == "cuda"

    def _ctx(self):
        import contextlib
        return _torch().cuda.device(self.eng.device) if self._on_gpu() else contextlib.nullcontext()

    def nbytes(self):
        return sum(int(b["t"].numel()) for b in self.bases)

    def idle(self, b):
        return _storage_users(b["t"]) <= 1

    def cap_bytes(self):
        env = os.environ.get("CAMA_POOL_BYTES")
        if env:
            return int(float(env))
        if not self._on_gpu():
            return 1 << 62
        total = _torch().cuda.get_device_properties(self.eng.device).total_memory
        return total // 8                   # idle bases beyond this are dropped least recently used first (live ones never)

    def trim(self, keep_bytes=0):
        """Drop idle bases, least recently used first, until at most `keep_bytes` are pooled (0: every idle one)."""
        n = 0
        for b in sorted(self.bases, key=lambda b: b["lru"]):
            if self.nbytes() <= keep_bytes:
                break
            if self.idle(b):
                self.bases = [x for x in self.bases if x is not b]      # (by identity: dict equality would compare tensors)
                n += 1
        self.stats["trimmed"] += n
        return n

    def _lend(self, b, F):
        torch = _torch()
        self.clock += 1
        b["lru"] = self.clock
        if not self._on_gpu():
            return b["t"][:F]
        cur = torch.cuda.current_stream(self.eng.device)
        if b["stream"] is not None and b["stream"].cuda_stream != cur.cuda_stream:
            # last used under another torch stream: whatever was queued there so far comes first
            ev = torch.cuda.Event()
            ev.record(b["stream"])
            cur.wait_event(ev)
        b["stream"] = cur
        return b["t"][:F]

    def _find_idle(self, frame, F):
        best = None
        for b in self.bases:
            Fb = int(b["t"].shape[0])
            # a longer base serves a shorter request as long as not much of it lies fallow
            if b["frame"] == frame and F <= Fb <= max(F + 8, F + F // 2) and self.idle(b):
                if best is None or (b["ms"] or 9e9, Fb) < (best["ms"] or 9e9, int(best["t"].shape[0])):
                    best = b
        return best

    def _alloc(self, shape):
        torch = _torch()
        try:
            return torch.empty(shape, dtype=torch.uint8, device=self.eng.device)
        except torch.OutOfMemoryError:
            if not self.trim(0):
                raise
            if self._on_gpu():
                torch.cuda.empty_cache()
            return torch.empty(shape, dtype=torch.uint8, device=self.eng.device)

    def _add(self, t, ms=None):
        b = {"t": t, "frame": tuple(int(v) for v in t.shape[1:]), "ms": ms, "stream": None, "lru": 0}
        self.bases.append(b)
        over = self.nbytes() - self.cap_bytes()
        if over > 0:
            self.trim(self.cap_bytes())
        return b

    def audition_candidates(self):
        return max(0, int(os.environ.get("CAMA_AUDITION", "16")))

    # A box either has a fast kind of (source, destination) placement -- the stamp-free overlay at >= 0.82 of 8 TB/s against <= 0.78
    # for the usual kind -- or it does not (round 5, the driver's box: 72 candidates within 2.1 % of each other, none fast; the
    # audition cost 133 GB of transient allocations and bought nothing).  So candidates are timed FOUR at a time and the audition
    # stops as soon as it knows: a fast one has been seen (keep it), or FLAT_AFTER candidates lie within FLAT_RATIO of each other
    # (keep candidate 0 -- exactly what a plain allocation would have been -- and never audition on this box again), or
    # CAMA_AUDITION (16) candidates / half of the free memory are used up (keep the fastest).
    FLAT_AFTER = 8
    FLAT_RATIO = 1.03
    GOOD_FRAC = 0.805
    KEEP = 2                            # bases kept per audition (the winner + runners-up within 3 %: a pipelined caller ping-pongs)

    def _audition(self, shape, n_keep, time_one, K):
        """Time candidate allocations of `shape` (time_one(tensor) -> ms) under the rules above.
        -> (candidates, times, verdict, seconds, peak transient bytes)."""
        import time as _time
        torch = _torch()
        nbytes = int(np.prod(shape))
        good_ms = 2.0 * nbytes / (self.GOOD_FRAC * 8.0e12) * 1e3
        t0 = _time.perf_counter()
        cands, times, verdict = [], [], "budget"
        while len(cands) < K:
            for _ in range(min(4, K - len(cands))):
                try:
                    c = torch.empty(shape, dtype=torch.uint8, device=self.eng.device)
                except torch.OutOfMemoryError:
                    K = len(cands)
                    break
                cands.append(c)                                       # alive together: distinct memory
                times.append(time_one(c))
            if not times:
                break
            fast = sum(1 for t in times if t <= good_ms)
            if fast >= n_keep:
                verdict = "fast placement found"
                break
            if len(times) >= max(self.FLAT_AFTER, n_keep) and not fast and max(times) < self.FLAT_RATIO * min(times):
                verdict = "no fast mode on this box"
                break
        if verdict == "no fast mode on this box":
            self.flat_box = True
        return cands, times, verdict, _time.perf_counter() - t0, len(cands) * nbytes

    # ---- the entry points
    def take(self, shape, rig=None, src=None, cols=3):
        """A mosaic tensor of `shape` = (F, rows*H, cols*W, 3): a view of an idle pooled base, else of a new one (placed
        against `src` [F,C,H,W,3] when it is worth it: see the class comment)."""
        shape = tuple(int(v) for v in shape)
        F, frame = shape[0], shape[1:]
        self.stats["takes"] += 1
        with self._ctx():
            b = self._find_idle(frame, F)
            if b is not None:
                self.stats["hits"] += 1
                return self._lend(b, F)
            nbytes = int(np.prod(shape))
            K = self.audition_candidates() if self._on_gpu() else 0
            eng = self.eng
            if (K <= 1 or self.flat_box or rig is None or src is None or nbytes < (1 << 29) or nbytes > (8 << 30) or F == 0
                    or int(src.shape[0]) != F or not eng._probeable(rig, src)):
                self.stats["allocations"] += 1
                return self._lend(self._add(self._alloc(shape)), F)
            free, _ = _torch().cuda.mem_get_info(eng.device)
            K = max(1, min(K, int(free // 2 // nbytes)))
            cands, times, verdict, secs, peak = self._audition(shape, 1, lambda c: eng._overlay_ms(rig, src, c, cols, 3), K)
            if not cands:
                self.stats["allocations"] += 1
                return self._lend(self._add(self._alloc(shape)), F)
            if self.flat_box:
                keep = [0]                                            # what a plain allocation would have been
            else:
                rank = sorted(range(len(cands)), key=lambda i: times[i])
                keep = [rank[0]] + [i for i in rank[1:self.KEEP] if times[i] <= 1.03 * times[rank[0]]]
            self.stats["auditions"] += 1
            self.stats["allocations"] += len(keep)
            self.stats["audition_seconds"] += secs
            self.stats["audition_peak_bytes"] = max(self.stats["audition_peak_bytes"], peak)
            eng._log_audition({"role": "mosaic", "bytes": nbytes, "candidates": len(cands), "ms": [round(t, 4) for t in times],
                               "chosen_ms": round(times[keep[0]], 4), "kept": len(keep), "source": "engine pool",
                               "verdict": verdict, "seconds": round(secs, 4), "peak_bytes": peak})
            kept = [self._add(cands[i], times[i]) for i in keep]
            del cands
            _torch().cuda.empty_cache()                               # the losers go back to the driver, not to torch's cache
            eng.settle_mapping(rig, src, kept[0]["t"], cols)
            return self._lend(kept[0], F)

    def take_many(self, shapes, rig, srcs, cols=3):
        """One mosaic per (shape, source) -- the scenes of a multi-scene launch chain, the launches of a long clip -- placed
        as the fastest of ONE pool of candidates (timing candidates one buffer at a time would mostly re-time the previous
        buffer's losers, which the allocator hands straight back).  Idle pooled bases are used first."""
        torch = _torch()
        shapes = [tuple(int(v) for v in sh) for sh in shapes]
        out = [None] * len(shapes)
        with self._ctx():
            for k, sh in enumerate(shapes):                            # what the pool already has
                b = self._find_idle(sh[1:], sh[0])
                if b is not None:
                    self.stats["takes"] += 1
                    self.stats["hits"] += 1
                    out[k] = self._lend(b, sh[0])                      # (lent: no longer idle for the next k)
            todo = [k for k in range(len(shapes)) if out[k] is None]
            if not todo:
                return out
            eng = self.eng
            Fmax = max(shapes[k][0] for k in todo)
            big = (Fmax,) + shapes[todo[0]][1:]
            nbytes = int(np.prod(big))
            n = len(todo)
            K = self.audition_candidates() if self._on_gpu() else 0
            P = 0
            if K > 1 and not self.flat_box and nbytes >= (1 << 29) and all(shapes[k][1:] == big[1:] for k in todo) \
                    and all(srcs[k] is not None and eng._probeable(rig, srcs[k]) for k in todo):
                free, _ = torch.cuda.mem_get_info(eng.device)
                # n buffers are needed anyway; beyond them at most K spare candidates (round 6: a pool of 4 per buffer timed 164
                # candidates = 170 GB of transient allocations for the 73-scene sweep on one GPU)
                P = min(n + K, int(free * 3 // 4 // nbytes))
            cands, times = [], []
            if P > n:
                k0 = todo[0]
                F0 = shapes[k0][0]
                cands, times, verdict, secs, peak = self._audition(
                    big, n, lambda c: eng._overlay_ms(rig, srcs[k0], c[:F0], cols, 3), P)
                self.stats["auditions"] += 1
                self.stats["audition_seconds"] += secs
                self.stats["audition_peak_bytes"] = max(self.stats["audition_peak_bytes"], peak)
                # flat box: the candidates in the order they came (= plain allocations); else fastest first
                rank = list(range(len(cands))) if self.flat_box else sorted(range(len(cands)), key=lambda i: times[i])
                rank = rank[:n]
                eng._log_audition({"role": "mosaic", "bytes": nbytes, "candidates": len(cands), "ms": [round(t, 4) for t in times],
                                   "chosen_ms": round(float(np.mean([times[i] for i in rank])), 4) if rank else None,
                                   "kept": len(rank), "source": "engine pool", "verdict": verdict, "seconds": round(secs, 4),
                                   "peak_bytes": peak})
                for i, k in zip(rank, todo):
                    self.stats["takes"] += 1
                    self.stats["allocations"] += 1
                    out[k] = self._lend(self._add(cands[i], times[i]), shapes[k][0])
                del cands
                torch.cuda.empty_cache()                               # the losers go back to the driver, not to torch's cache
            for k in todo:                                             # no audition, or fewer candidates than buffers
                if out[k] is None:
                    self.stats["takes"] += 1
                    self.stats["allocations"] += 1
                    out[k] = self._lend(self._add(self._alloc(shapes[k])), shapes[k][0])
            return out


class Engine:
    def __init__(self, device="cuda:0", crop=CROP_BOX, radius=RADIUS, palette_bgr=PALETTE_BGR, alpha=1.0):
        torch = _torch()
        self.lib = _lib.lib()                                   # raises if the .so is missing
        if not torch.cuda.is_available():
            raise _lib.CamaHipError("cama_amd.Engine needs a GPU (torch.cuda.is_available() is False); "
                                    "there is no CPU fallback")
        self.device = torch.device(device)
        self.crop = np.asarray(crop, np.float64)
        self.radius = int(radius)
        # extension: translucent stamps (the reference is opaque = 1.0); applied by render_frames only
        self.alpha256 = int(round(float(alpha) * 256))
        assert 0 <= self.alpha256 <= 256
        self.halfwidth = _lib.circle_halfwidths(self.radius)
        self.palette = np.ascontiguousarray(np.asarray(palette_bgr, np.uint8).reshape(2, 3))
        self._scratch = None
        self._pipe = None
        self.pool = MosaicPool(self)

    # ------------------------------------------------------------------ process-wide ingest state (one Engine per GPU)
    # A ClipManager lives for one scene (main.py:50); what its frame source needs to turn files into device frames does not
    # have to: the JPEG decoder (its lanes = streams + pinned staging + device scratch, its pinned read arenas), the reader
    # threads and the decode pump's stream are owned HERE and shared by every ClipFrameSource on this GPU.  Round 5,
    # profiles/r05_cold_sweep.txt: per-clip copies of these cost ~100 ms of every scene's first frame (hipHostMalloc of the
    # arenas, thread start-up) and 3.2 GB of new device segments per scene -- blocks allocated under a clip's own pump
    # stream sit in that stream's pool of torch's caching allocator, where the next clip's stream cannot reuse them.
    def jpeg_decoder(self):
        dec = self.__dict__.get("_jpeg_decoder")
        if dec is None:
            from .jpeg import DeviceJpegDecoder
            dec = self.__dict__["_jpeg_decoder"] = DeviceJpegDecoder(self.device)
        return dec

    def reader_pool(self, workers):
        """The shared file-reader thread pool (grown, never shrunk: `workers` is the most any source asked for)."""
        from concurrent.futures import ThreadPoolExecutor
        pool = self.__dict__.get("_reader_pool")
        if pool is None or pool._max_workers < workers:
            pool = self.__dict__["_reader_pool"] = ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix="cama-read")
        return pool

    def pump_stream(self):
        st = self.__dict__.get("_pump_stream")
        if st is None:
            st = self.__dict__["_pump_stream"] = _torch().cuda.Stream(device=self.device)
        return st

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return _torch().cuda.current_stream(self.device).cuda_stream

    def _scratch_buf(self, nbytes):
        torch = _torch()
        if self._scratch is None or self._scratch.numel() < nbytes:
            self._scratch = None
            self._scratch = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._scratch

    def _mats(self, mats):
        """(F,4,4) array-like (float32 or float64, host) -> device [F,16] float64 (exact promotion)."""
        torch = _torch()
        if isinstance(mats, torch.Tensor):
            m = mats.to(device=self.device, dtype=torch.float64).reshape(-1, 16).contiguous()
            return m
        m = np.ascontiguousarray(np.asarray(mats, dtype=np.float64).reshape(-1, 16))
        return torch.from_numpy(m).to(self.device, non_blocking=True)

    def build_static_map(self, table, lift, bev_height=None, solution=0.1, map_width=600, map_height=600,
                         center_x=0, center_y=0):
        """Device static-map build (cama_build_static_map) from MapManager.segment_table(): returns a DeviceMap whose
        vertex buffer was produced on the GPU (bit-identical to the host build); nothing is copied back."""
        torch = _torch()
        with torch.cuda.device(self.device):
            N = int(table["seg_off"][-1])
            S = int(table["seg_num"].shape[0])
            # The raster's dtype decides the vertex dtype: the reference concatenates float32 xy with the gathered
            # heights (cama/reproject.py:96-103), i.e. numpy promotion of (float32, raster dtype) -- float32 for
            # float16 / int8 / int16 / uint8 / uint16 rasters, float64 for float64 / int32 / int64 ones.  Normalise the
            # raster to that type BEFORE anything is sized from it (every such conversion is exact).
            bev = None
            if lift:
                bev = np.asarray(bev_height)
                assert bev.ndim == 2
                res = np.result_type(np.float32, bev.dtype)
                bev = bev.astype(np.float64 if res == np.float64 else np.float32, copy=False)
            is64 = bool(lift and bev.dtype == np.float64)
            dt = torch.float64 if is64 else torch.float32
            soa = torch.empty((3, N), dtype=dt, device=self.device)
            colour = torch.empty((N,), dtype=torch.uint8, device=self.device)
            dmap = DeviceMap.__new__(DeviceMap)
            dmap.N, dmap.is_f64, dmap.soa, dmap.colour = N, int(is64), soa, colour
            dmap.sorted_soa = dmap.sorted_key = None
            dmap.has_links = False
            dmap.bounds, dmap.extent_xy = None, (0.0, 0.0)
            if N == 0:
                return dmap
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            verts, v0, num, off, col = (up(table[k]) for k in ("verts", "seg_v0", "seg_num", "seg_off", "seg_colour"))
            raster = None
            rows = cols = 0
            if lift:
                rows, cols = int(bev.shape[0]), int(bev.shape[1])
                raster = up(bev)
            es = soa.element_size()
            base = soa.data_ptr()
            f32 = lambda v: float(np.float32(v))
            _lib.check(self.lib.cama_build_static_map(
                verts.data_ptr(), v0.data_ptr(), num.data_ptr(), off.data_ptr(), col.data_ptr(), S, N, int(bool(lift)),
                None if raster is None else raster.data_ptr(), int(is64), rows, cols,
                f32(solution), f32(map_width / 2), f32(map_height / 2), f32(center_x), f32(center_y),
                base, base + N * es, base + 2 * N * es, colour.data_ptr(), self._stream()))
            dmap._index()                                             # synchronises: the uploads above may go now
            torch.cuda.current_stream(self.device).synchronize()
            return dmap

    def upload_map(self, xyz, colour_id, spatial_sort="auto"):
        return DeviceMap(xyz, colour_id, self.device, spatial_sort=spatial_sort)

    def shared_map(self, xyz, colour_id, spatial_sort="auto"):
        """Content-keyed DeviceMap: clips whose static maps are the same bytes -- CAMA v2's site-aggregated labels: every
        scene of a site carries the site's labels (main.py:42, cama/reproject.py:26-27) -- share ONE device copy on this
        GPU: one upload, one Morton sort, one block index per site, however many ClipManagers ask.  Weak: a map goes when
        its last clip goes.  `map_cache_stats` counts uploads and hits."""
        import weakref
        xyz = np.asarray(xyz)
        if xyz.dtype not in (np.float32, np.float64):
            xyz = xyz.astype(np.float64)
        col = np.ascontiguousarray(colour_id, dtype=np.uint8)
        h = _content_hasher()
        h.update(np.ascontiguousarray(xyz).view(np.uint8).reshape(-1).data)
        h.update(col.data)
        key = (h.hexdigest(), xyz.shape, str(xyz.dtype), str(spatial_sort))
        cache = self.__dict__.setdefault("_map_cache", weakref.WeakValueDictionary())
        stats = self.__dict__.setdefault("map_cache_stats", {"uploads": 0, "hits": 0})
        dmap = cache.get(key)
        if dmap is None:
            dmap = DeviceMap(xyz, col, self.device, spatial_sort=spatial_sort)
            cache[key] = dmap
            stats["uploads"] += 1
        else:
            stats["hits"] += 1
        return dmap

    def make_rig(self, names, chassis2camera, K, W, H):
        return CameraRig(names, chassis2camera, K, W, H, self.device)

    # ------------------------------------------------------------------ API-mode kernels
    def transform_points(self, xyz, mats, crop=None):
        """(N,3) points x F matrices -> (out [F,N,3] float64, mask [F,N] uint8) device tensors."""
        torch = _torch()
        with torch.cuda.device(self.device):
            if isinstance(xyz, torch.Tensor):
                p = xyz.to(self.device).contiguous()
                is64 = int(p.dtype == torch.float64)
                assert p.dtype in (torch.float32, torch.float64)
            else:
                a = np.asarray(xyz)
                if a.dtype not in (np.float32, np.float64):
                    a = a.astype(np.float64)
                is64 = int(a.dtype == np.float64)
                p = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            N = p.shape[0]
            T = self._mats(mats)
            F = T.shape[0]
            out = torch.empty((F, N, 3), dtype=torch.float64, device=self.device)
            mask = torch.empty((F, N), dtype=torch.uint8, device=self.device)
            cropa = None if crop is None else np.ascontiguousarray(np.asarray(crop, np.float64))
            _lib.check(self.lib.cama_transform_points(
                p.data_ptr(), is64, N, T.data_ptr(), F, None if cropa is None else cropa.ctypes.data,
                out.data_ptr(), mask.data_ptr(), self._stream()))
            return out, mask

    def crop_points(self, xyz, crop):
        """(n,3) float64 points -> mask [n] uint8 (inclusive box test on the device)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            p = torch.from_numpy(np.ascontiguousarray(np.asarray(xyz, np.float64))).to(self.device)
            mask = torch.empty((p.shape[0],), dtype=torch.uint8, device=self.device)
            cropa = np.ascontiguousarray(np.asarray(crop, np.float64))
            _lib.check(self.lib.cama_crop_points(p.data_ptr(), p.shape[0], cropa.ctypes.data, mask.data_ptr(),
                                                 self._stream()))
            return mask

    def project_points(self, rig, chassis_xyz):
        """(n,3) float64 chassis-frame points -> (vu [C,n,2] float64, vis [C,n] uint8)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            if isinstance(chassis_xyz, torch.Tensor):
                p = chassis_xyz.to(device=self.device, dtype=torch.float64).contiguous()
            else:
                p = torch.from_numpy(np.ascontiguousarray(np.asarray(chassis_xyz, np.float64))).to(self.device)
            n = p.shape[0]
            vu = torch.empty((rig.C, n, 2), dtype=torch.float64, device=self.device)
            vis = torch.empty((rig.C, n), dtype=torch.uint8, device=self.device)
            _lib.check(self.lib.cama_project_points(p.data_ptr(), n, rig.c2cam.data_ptr(), rig.K.data_ptr(), rig.C,
                                                    rig.W, rig.H, vu.data_ptr(), vis.data_ptr(), self._stream()))
            return vu, vis

    def _crop(self, crop):
        return self.crop if crop is None else np.ascontiguousarray(np.asarray(crop, np.float64))

    def project_frames(self, dmap, rig, w2c, crop=None):
        """Fused chain, coordinates materialised: (vu [F,C,N,2], vis [F,C,N], crop_mask [F,N])."""
        torch = _torch()
        cropa = self._crop(crop)
        with torch.cuda.device(self.device):
            T = self._mats(w2c)
            F = T.shape[0]
            vu = torch.full((F, rig.C, dmap.N, 2), float("nan"), dtype=torch.float64, device=self.device)
            vis = torch.empty((F, rig.C, dmap.N), dtype=torch.uint8, device=self.device)
            cm = torch.empty((F, dmap.N), dtype=torch.uint8, device=self.device)
            x, y, z = dmap.ptrs()
            _lib.check(self.lib.cama_project_frames(
                x, y, z, dmap.is_f64, dmap.N, T.data_ptr(), F, rig.c2cam.data_ptr(), rig.K.data_ptr(), rig.C,
                cropa.ctypes.data, rig.W, rig.H, vu.data_ptr(), vis.data_ptr(), cm.data_ptr(), self._stream()))
            return vu, vis, cm

    # ------------------------------------------------------------------ fused render
    def mosaic_shape(self, rig, F, cols=3):
        rows = (rig.C + cols - 1) // cols
        return (F, rows * rig.H, cols * rig.W, 3)

    def render_frames(self, dmap, rig, w2c, src, out=None, cols=3, crop=None, segments=False):
        """src [F,C,H,W,3] uint8 device tensor -> mosaic [F, rows*H, cols*W, 3] uint8 device tensor."""
        torch = _torch()
        cropa = self._crop(crop)
        with torch.cuda.device(self.device):
            T = w2c if (isinstance(w2c, torch.Tensor) and w2c.dtype == torch.float64 and w2c.is_cuda
                        and w2c.dim() == 2) else self._mats(w2c)
            F = T.shape[0]
            assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
            assert tuple(src.shape) == (F, rig.C, rig.H, rig.W, 3), (tuple(src.shape), (F, rig.C, rig.H, rig.W, 3))
            shape = self.mosaic_shape(rig, F, cols)
            if out is None:                 # a view of one of the engine's pooled, placed buffers (MosaicPool)
                out = self.pool.take(shape, rig, src, cols)
            assert tuple(out.shape) == shape and out.is_contiguous() and out.dtype == torch.uint8
            x, y, z, col, key, bnd, bflags = dmap.render_ptrs(cropa)
            if segments:                    # extension: only through the pipeline (its sorted list is sized per launch)
                self.render_frames_pipelined(dmap, rig, T, src, out, cols=cols, crop=crop, segments=segments)
                self.join()
                return out
            if self.alpha256 == 256 and bnd is not None and (bflags & _lib.BIN_WORKLIST):
                # site-sized map: the caller-scratch entry point below would size its stamp scratch for the worst case
                # (24 B per frame x camera x vertex); the pipeline plans the launch and sizes it from what survives the
                # cull (cama_pipeline_render with its own scratch).  Same kernels; complete on the current stream on return.
                self.render_frames_pipelined(dmap, rig, T, src, out, cols=cols, crop=crop)
                self.join()
                return out
            need = self.lib.cama_render_scratch_bytes(dmap.N, F, rig.C, rig.H, rig.W, self.radius)
            scratch = self._scratch_buf(need)
            if self.alpha256 != 256:        # extension path: binning + translucent overlay
                _lib.check(self.lib.cama_bin_frames(
                    x, y, z, dmap.is_f64, col, key, bnd, bflags, dmap.N, T.data_ptr(), F, rig.c2cam.data_ptr(), rig.K.data_ptr(),
                    rig.C, cropa.ctypes.data, rig.W, rig.H, self.radius, scratch.data_ptr(), scratch.numel(),
                    self._stream()))
                _lib.check(self.lib.cama_overlay_frames_alpha(
                    src.data_ptr(), out.data_ptr(), dmap.N, F, rig.C, rig.H, rig.W, cols, self.radius,
                    self.halfwidth.ctypes.data, self.palette.ctypes.data, self.alpha256, scratch.data_ptr(),
                    scratch.numel(), self._stream()))
                return out
            _lib.check(self.lib.cama_render_frames(
                x, y, z, dmap.is_f64, col, key, bnd, bflags, dmap.N, T.data_ptr(), F,
                rig.c2cam.data_ptr(), rig.K.data_ptr(), rig.C, cropa.ctypes.data, rig.W, rig.H,
                src.data_ptr(), out.data_ptr(), cols, self.radius, self.halfwidth.ctypes.data,
                self.palette.ctypes.data, scratch.data_ptr(), scratch.numel(), self._stream()))
            self._last_bin = (dmap.N, F, rig.C, rig.H, rig.W, bnd is not None, key is not None, scratch)
            return out

    # ------------------------------------------------------------------ placement of long-lived buffers (mosaics: MosaicPool above)
    def _log_audition(self, entry):
        """What the placement helpers saw (bench.py reports it): the most recent 256 entries."""
        log = self.__dict__.setdefault("audition_log", [])
        log.append(entry)
        del log[:-256]

    def _probeable(self, rig, src):
        """Can cama_overlay_probe time launches from `src`?  (contiguous uint8 [F, C, H, W, 3] of the rig's size, W % 16 == 0,
        opaque stamps)"""
        return (src.dim() == 5 and tuple(src.shape[1:]) == (rig.C, rig.H, rig.W, 3) and src.is_contiguous()
                and str(src.dtype) == "torch.uint8" and src.data_ptr() % 16 == 0 and rig.W % 16 == 0 and self.alpha256 == 256)

    def _overlay_ms(self, rig, src, out, cols, reps):
        """Mean duration (ms) of `reps` stamp-free overlay launches src -> out in the XCD-contiguous order
        (cama_overlay_probe: its own kernel name, not through the mapping table; blocks)."""
        import ctypes
        ms = ctypes.c_double(0.0)
        _lib.check(self.lib.cama_overlay_probe(src.data_ptr(), out.data_ptr(), int(src.shape[0]), rig.C, rig.H, rig.W, cols,
                                               reps, ctypes.byref(ms), self._stream()))
        return ms.value

    def settle_mapping(self, rig, src, out, cols=3):
        """Let the library finish choosing the workgroup -> band order for the pair (src, out) NOW -- it times both orders
        on the first six big launches over a pair (MapTuner, cama_hip.hip) -- with stamp-free launches, so that none of the
        caller's own launches over this pair is a trial.  No-op for launches below the tuner's 1.75 GiB."""
        torch = _torch()
        F = int(src.shape[0])
        if 6 * F * rig.C * rig.H * rig.W < (7 << 28) or not self._probeable(rig, src):
            return
        L = self.lib
        with torch.cuda.device(self.device):
            need = int(L.cama_render_scratch_bytes(0, F, rig.C, rig.H, rig.W, self.radius))
            scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
            w2c = torch.zeros((F, 16), dtype=torch.float64, device=self.device)
            _lib.check(L.cama_bin_frames(None, None, None, 0, None, None, None, 0, 0, w2c.data_ptr(), F, rig.c2cam.data_ptr(),
                                         rig.K.data_ptr(), rig.C, self.crop.ctypes.data, rig.W, rig.H, self.radius,
                                         scratch.data_ptr(), scratch.numel(), self._stream()))
            for _ in range(4 * 3 + 1):
                _lib.check(L.cama_overlay_frames(src.data_ptr(), out.data_ptr(), 0, F, rig.C, rig.H, rig.W, cols, self.radius,
                                                 self.halfwidth.ctypes.data, self.palette.ctypes.data, scratch.data_ptr(),
                                                 scratch.numel(), self._stream()))
                torch.cuda.current_stream(self.device).synchronize()
                if self.overlay_mapping()["decided"] >= 0:
                    break

    def place_frames(self, rig, frames, out, first=0, cols=3, candidates=None, reps=3):
        """The counterpart for the SOURCE side: `frames` [>= first + F, C,H,W,3] (already filled; the F = out.shape[0] frames
        from index `first` are what gets rendered) is copied into the one of `candidates` fresh buffers from which the overlay
        into `out` runs fastest (the source's placement is worth 2-4 % once the mosaic's is good).  Returns the buffer to use
        from now on (possibly `frames` itself)."""
        torch = _torch()
        F = int(out.shape[0])
        view = lambda t: t[first:first + F]
        nbytes = int(frames.numel())
        K = int(os.environ.get("CAMA_AUDITION", "16")) // 2 if candidates is None else int(candidates)
        with torch.cuda.device(self.device):
            if (K <= 1 or self.pool.flat_box or nbytes < (1 << 29) or nbytes > (8 << 30)
                    or not self._probeable(rig, view(frames))):
                return frames                                      # (a box without a fast placement has none for the source either)
            free, _ = torch.cuda.mem_get_info(self.device)
            K = max(1, min(K, int(free // 2 // nbytes)))
            import time as _time
            t0 = _time.perf_counter()
            best, best_ms = frames, self._overlay_ms(rig, view(frames), out, cols, reps)
            times, pool = [best_ms], []
            for _ in range(K):
                cand = torch.empty_like(frames)
                pool.append(cand)
                ms = self._overlay_ms(rig, view(cand), out, cols, reps)     # (content does not matter for the timing)
                times.append(ms)
                if ms < best_ms * 0.995:
                    best, best_ms = cand, ms
            if best is not frames:
                best.copy_(frames)
                del pool
                self.settle_mapping(rig, view(best), out, cols)
            secs = _time.perf_counter() - t0
            self.pool.stats["audition_seconds"] += secs
            self.pool.stats["audition_peak_bytes"] = max(self.pool.stats["audition_peak_bytes"], K * nbytes)
            self._log_audition(
                {"role": "frames", "bytes": nbytes, "candidates": K, "ms": [round(t, 4) for t in times], "chosen_ms": round(best_ms, 4),
                 "moved": best is not frames, "seconds": round(secs, 4), "peak_bytes": K * nbytes})
            return best

    def xcd_map(self, n_blocks=4096):
        """What the overlay's XCD-contiguous mapping relies on, measured: (True when the XCD a block of a 1-D grid runs on
        is a function of its index mod 8 -- round-robin from whichever XCD the dispatcher is at: then every XCD gets every
        eighth block and hence one contiguous eighth of the bands --, the XCDs of blocks 0..7, blocks per XCD)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            out = torch.empty(n_blocks, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.cama_probe_xcd_map(out.data_ptr(), n_blocks, self._stream()))
            got = out.cpu().numpy()
        periodic = all(len(set(got[k::8].tolist())) == 1 for k in range(8)) and len(set(got[:8].tolist())) == 8
        return bool(periodic), [int(v) for v in got[:8]], np.bincount(got, minlength=8).tolist()

    def overlay_mapping(self):
        """Which workgroup -> band order this process's big overlay launches use (cama_overlay_mapping_info): dict with
        decided (-1 while the library is still timing both, else 31 = contiguous per XCD or 5 = round-robin chunks of 32 bands),
        samples and best ns per 10^6 bytes for [contiguous, chunked]."""
        import ctypes
        d = ctypes.c_int32(-1)
        n = (ctypes.c_int32 * 2)()
        t = (ctypes.c_double * 2)()
        _lib.check(self.lib.cama_overlay_mapping_info(ctypes.byref(d), n, t))
        return {"decided": int(d.value), "samples": [int(n[0]), int(n[1])], "ns_per_mb": [float(t[0]), float(t[1])]}

    def bin_stats(self):
        """What the binning half of the LAST render_frames() call read and produced (cama_bin_stats; blocks until the
        stream is idle): dict with frames, vertex_waves_read (64-vertex runs fetched, over all frames), camera_chains,
        stamps, band_entries, vertex_bytes_read (13 B per vertex, 16 B when the map carries a draw key), or None."""
        last = getattr(self, "_last_bin", None)
        if last is None:
            return None
        N, F, C, H, W, had_bounds, has_key, scratch = last
        out = np.zeros(4, np.uint64)
        with _torch().cuda.device(self.device):
            if scratch is None:             # the launch went through the pipeline's own scratch
                if self._pipe is None:      # ... which shrink_frames_per_call() has since destroyed: nothing to read
                    return None
                _lib.check(self.lib.cama_pipeline_bin_stats(self._pipe["handle"], out.ctypes.data))
            else:
                _lib.check(self.lib.cama_bin_stats(scratch.data_ptr(), scratch.numel(), N, F, C, H, W, self.radius,
                                                   int(had_bounds), out.ctypes.data, self._stream()))
        per_vertex = 16 if has_key else 13
        return {"frames": F, "verts": N, "vertex_waves_read": int(out[0]), "camera_chains": int(out[1]),
                "stamps": int(out[2]), "band_entries": int(out[3]),
                "vertex_bytes_read": int(min(int(out[0]) * 64, N * F)) * per_vertex,
                "block_cull": bool(had_bounds)}

    # ------------------------------------------------------------------ frame resample (undistort + resize)
    def resample(self, cm, src, out=None):
        """src [n,H0,W0,3] (or [H0,W0,3]) uint8 device frames of CameraManager `cm` -> [n,H,W,3] at its output size.
        `out` may be a strided view whose frames are each contiguous (e.g. mosaic-source[:, c])."""
        torch = _torch()
        from .frames import camera_maps_compact
        with torch.cuda.device(self.device):
            single = src.dim() == 3
            s = src[None] if single else src
            assert s.is_cuda and s.dtype == torch.uint8 and s[0].is_contiguous() and s.shape[3] == 3
            n, H0, W0 = int(s.shape[0]), int(s.shape[1]), int(s.shape[2])
            src_stride = s.stride(0) if n > 1 else H0 * W0 * 3
            H, W = int(cm.height), int(cm.width)
            dev_maps = getattr(cm, "_resample_maps_dev", None)
            if dev_maps is None or dev_maps[0].device != self.device:
                # zero distortion gives separable maps: a W-vector and an H-vector instead of two H x W planes
                mx, my, sep = camera_maps_compact(cm)
                dev_maps = (torch.from_numpy(mx).to(self.device), torch.from_numpy(my).to(self.device), int(sep))
                cm._resample_maps_dev = dev_maps
            if out is None:
                out = torch.empty((n, H, W, 3), dtype=torch.uint8, device=self.device)
            assert tuple(out.shape) == (n, H, W, 3) and out.dtype == torch.uint8 and out[0].is_contiguous()
            stride = out.stride(0) if n > 1 else H * W * 3
            _lib.check(self.lib.cama_resample_frames(s.data_ptr(), src_stride, out.data_ptr(), stride, n, H0, W0, H, W,
                                                     dev_maps[0].data_ptr(), dev_maps[1].data_ptr(), dev_maps[2],
                                                     self._stream()))
            return out[0] if single else out

    # ------------------------------------------------------------------ fused render from RAW sensor frames
    def rig_maps(self, cm_list):
        """Per-camera undistort/resize maps of a rig, concatenated on the device: (mapx, mapy, separable)."""
        torch = _torch()
        from .frames import camera_maps_compact
        # keyed on CONTENT (sizes, both intrinsics, distortion), not on object identity: ids are recycled, and two clips
        # of one rig share a plan.  A small dict of plans: the tensors of a plan that queued launches still read stay
        # referenced from the launches' keep tuples (render_frames_raw) as well.
        def _sig(cm):
            d = cm.d_origin if cm.d == [] else cm.d
            return (int(cm.height), int(cm.width), int(cm.height_origin), int(cm.width_origin),
                    np.asarray(cm.K_origin, np.float64).tobytes(), np.asarray(cm.K, np.float64).tobytes(),
                    b"" if d is None else np.asarray(d, np.float64).tobytes())
        key = tuple(_sig(cm) for cm in cm_list)
        plans = self.__dict__.setdefault("_rig_plans", {})
        hit = plans.get(key)
        if hit is None:
            maps = [camera_maps_compact(cm) for cm in cm_list]
            sep = all(m[2] for m in maps)
            if sep:
                mx = np.stack([m[0] for m in maps])
                my = np.stack([m[1] for m in maps])
            else:                                   # (a rig with one distorted camera: planes for all of them)
                from .frames import camera_maps
                full = [camera_maps(cm) for cm in cm_list]
                mx = np.stack([m[0].reshape(-1) for m in full])
                my = np.stack([m[1].reshape(-1) for m in full])
            # rational 3:5 scale (the reference default 1600x900 -> 960x540): the library verifies the tap pattern and
            # builds the per-row vertical taps for its gather-free kernel (cama_raw35_plan)
            vrows = None
            if sep and USE_RAW35:
                Cn, Hd, Wd = len(cm_list), int(cm_list[0].height), int(cm_list[0].width)
                mxh = np.ascontiguousarray(mx, np.float32)
                myh = np.ascontiguousarray(my, np.float32)
                table = np.zeros((Cn, Hd, 2), np.uint32)
                R35 = self.lib.cama_overlay_band_rows(Wd)
                brows = np.zeros((Cn, (Hd + R35 - 1) // R35, 2), np.int32)
                import ctypes
                most = ctypes.c_int32(0)
                rc = self.lib.cama_raw35_plan(mxh.ctypes.data, myh.ctypes.data, Cn, Hd, Wd,
                                              int(cm_list[0].height_origin), int(cm_list[0].width_origin), table.ctypes.data,
                                              brows.ctypes.data, ctypes.byref(most))
                if rc < 0:
                    _lib.check(rc)
                if rc == 1:
                    vrows = (torch.from_numpy(table.view(np.int32)).to(self.device), torch.from_numpy(brows).to(self.device),
                             int(most.value))
            # (only for maps the 3:5 kernel does not take: its plan carries its own band table -- and building these for
            # every new calibration was 2-3 ms of a new clip's first frame)
            band_rows, max_rows, tiles, tiles_x, max_tile = None, 0, None, 0, 0
            if sep and vrows is None:
                # source rows every band of R destination rows touches (same rounding as the kernel: 1/32 px)
                H, H0 = int(cm_list[0].height), int(cm_list[0].height_origin)
                R = self.lib.cama_overlay_band_rows(int(cm_list[0].width))
                NB = (H + R - 1) // R
                yy0 = np.rint(my.astype(np.float32) * np.float32(32)).astype(np.int64) >> 5      # [C,H]
                if (np.diff(yy0, axis=1) >= 0).all():
                    rows = np.zeros((len(cm_list), NB, 2), np.int32)
                    for b in range(NB):
                        lo = np.clip(yy0[:, b * R], 0, H0 - 1)
                        hi = np.clip(yy0[:, min(H, b * R + R) - 1] + 1, 0, H0 - 1)
                        rows[:, b, 0] = lo
                        rows[:, b, 1] = hi - lo + 1
                    band_rows = torch.from_numpy(rows).to(self.device)
                    max_rows = int(rows[:, :, 1].max())
                    # column tiles of ~256-384 destination pixels (a multiple of 16): source byte range of each
                    Wd, W0 = int(cm_list[0].width), int(cm_list[0].width_origin)
                    tiles_x = max([t for t in range(1, Wd // 16 + 1)
                                   if Wd % t == 0 and (Wd // t) % 16 == 0 and Wd // t >= 256] or [1])
                    Wt = Wd // tiles_x
                    x0 = np.clip(np.rint(mx.astype(np.float32) * np.float32(32)).astype(np.int64) >> 5, 0, W0 - 1)  # [C,W]
                    tb = np.zeros((len(cm_list), tiles_x, 2), np.int32)
                    for t in range(tiles_x):
                        off = x0[:, t * Wt:(t + 1) * Wt] * 3
                        start = (off.min(axis=1) // 16) * 16
                        end = np.minimum(((off.max(axis=1) // 4) * 4 + 12 + 15) // 16 * 16, W0 * 3)
                        tb[:, t, 0] = start
                        tb[:, t, 1] = end - start
                    max_tile = int(tb[:, :, 1].max())
                    tb[:, :, 1] = max_tile                      # one LDS row stride for every tile (clamped below)
                    tb[:, :, 0] = np.minimum(tb[:, :, 0], W0 * 3 - max_tile)
                    tiles = torch.from_numpy(tb).to(self.device)
            hit = (key, torch.from_numpy(np.ascontiguousarray(mx)).to(self.device),
                   torch.from_numpy(np.ascontiguousarray(my)).to(self.device), int(sep), band_rows, max_rows,
                   tiles, tiles_x, max_tile, vrows)
            if len(plans) >= 16:                      # bounded: drop the oldest plan (dict order = insertion order)
                plans.pop(next(iter(plans)))
            plans[key] = hit
        return hit[1:]

    def render_frames_raw(self, dmap, rig, w2c, raw, cm_list, out=None, cols=3, crop=None, pipelined=False):
        """Like render_frames, but `raw` [F,C,H0,W0,3] holds RAW sensor frames: undistort + resize to the rig's
        output size happens inside the overlay kernel's source read (cama_overlay_frames_raw35 for the reference's
        3:5 scale, else cama_overlay_frames_raw).  pipelined=True (3:5 scale only; otherwise ignored) goes through the
        two-stream pipeline like render_frames_pipelined: `out` is complete after join()."""
        torch = _torch()
        cropa = self._crop(crop)
        with torch.cuda.device(self.device):
            host32 = pipelined and isinstance(w2c, np.ndarray) and w2c.dtype == np.float32      # staged by the pipeline
            if host32:
                T, F = None, int(w2c.reshape(-1, 16).shape[0])
            else:
                T = w2c if (isinstance(w2c, torch.Tensor) and w2c.dtype == torch.float64 and w2c.is_cuda
                            and w2c.dim() == 2) else self._mats(w2c)
                F = T.shape[0]
            assert raw.is_cuda and raw.dtype == torch.uint8 and raw.is_contiguous() and raw.dim() == 5
            assert raw.shape[0] == F and raw.shape[1] == rig.C and raw.shape[4] == 3
            H0, W0 = int(raw.shape[2]), int(raw.shape[3])
            shape = self.mosaic_shape(rig, F, cols)
            if out is None:                 # pooled (raw sources are not probed: a plain, recycled allocation)
                out = self.pool.take(shape)
            assert tuple(out.shape) == shape and out.is_contiguous()
            mapx, mapy, sep, band_rows, max_rows, tiles, tiles_x, max_tile, vrows = self.rig_maps(cm_list)
            x, y, z, col, key, bnd, bflags = dmap.render_ptrs(cropa)
            st = self._stream()
            if pipelined and vrows is not None:
                P = self._pipeline()
                staged = self._stage_poses(P, w2c)
                T_ptr = staged[0] if staged is not None else T.data_ptr()
                _lib.call_retrying_oom(self.lib.cama_pipeline_render_raw35,
                    P["handle"], x, y, z, dmap.is_f64, col, key, bnd, bflags, dmap.N, T_ptr, F, rig.c2cam.data_ptr(),
                    rig.K.data_ptr(), rig.C, cropa.ctypes.data, rig.W, rig.H, raw.data_ptr(), H0, W0, vrows[0].data_ptr(),
                    vrows[1].data_ptr(), vrows[2], out.data_ptr(), cols, self.radius, self.halfwidth.ctypes.data,
                    self.palette.ctypes.data, None, None, 0, st)       # (scratch: the pipeline's own, demand-sized)
                seq = int(self.lib.cama_pipeline_issued(P["handle"]))
                # the overlay runs later, on the pipeline's own stream, and reads the tap tables: they belong to the
                # launch's keep set like the frames and the map (a later call with another rig may evict the plan)
                P["keep"].append((seq, T, raw, out, dmap, rig, (vrows[0], vrows[1], mapx, mapy, band_rows, tiles)))
                self._release_completed(P)
                return out
            if T is None:
                T = self._mats(w2c)
            need = self.lib.cama_render_scratch_bytes(dmap.N, F, rig.C, rig.H, rig.W, self.radius)
            scratch = self._scratch_buf(need)
            _lib.check(self.lib.cama_bin_frames(
                x, y, z, dmap.is_f64, col, key, bnd, bflags, dmap.N, T.data_ptr(), F, rig.c2cam.data_ptr(), rig.K.data_ptr(), rig.C,
                cropa.ctypes.data, rig.W, rig.H, self.radius, scratch.data_ptr(), scratch.numel(), st))
            if vrows is not None:
                _lib.check(self.lib.cama_overlay_frames_raw35(
                    raw.data_ptr(), H0, W0, vrows[0].data_ptr(), vrows[1].data_ptr(), vrows[2], out.data_ptr(), dmap.N, F,
                    rig.C, rig.H, rig.W, cols,
                    self.radius, self.halfwidth.ctypes.data, self.palette.ctypes.data, scratch.data_ptr(),
                    scratch.numel(), st))
                return out
            _lib.check(self.lib.cama_overlay_frames_raw(
                raw.data_ptr(), H0, W0, mapx.data_ptr(), mapy.data_ptr(), sep,
                None if band_rows is None else band_rows.data_ptr(), max_rows,
                None if tiles is None else tiles.data_ptr(), tiles_x, max_tile, out.data_ptr(), dmap.N, F, rig.C,
                rig.H, rig.W, cols, self.radius, self.halfwidth.ctypes.data, self.palette.ctypes.data,
                scratch.data_ptr(), scratch.numel(), st))
            return out

    # ------------------------------------------------------------------ pipelined render (two streams)
    def _pipeline(self):
        import ctypes
        if self._pipe is None:
            handle = ctypes.c_void_p()
            with _torch().cuda.device(self.device):          # its streams live on this engine's GPU
                _lib.check(self.lib.cama_pipeline_create(ctypes.byref(handle)))
            self._pipe = {"handle": handle, "keep": []}
        return self._pipe

    def pipeline_info(self):
        """What the pipeline's own scratch looks like (cama_pipeline_info): dict with launches, planned_launches, grows,
        scratch_bytes, last_plan_segments, last_plan_capacity, tall_band_launches (launches that ran with 8-row bands: chosen per
        launch from the map's measured stamp density), last_band_rows; None before the first pipelined launch."""
        if self._pipe is None:
            return None
        out = np.zeros(8, np.uint64)
        _lib.check(self.lib.cama_pipeline_info(self._pipe["handle"], out.ctypes.data))
        return dict(zip(("launches", "planned_launches", "grows", "scratch_bytes", "last_plan_segments",
                         "last_plan_capacity", "tall_band_launches", "last_band_rows"), (int(v) for v in out)))

    def scratch_bytes(self):
        """Device bytes of stamp scratch this engine holds: the pipeline's own (demand-sized) + the single-stream buffer."""
        n = 0 if self._scratch is None else int(self._scratch.numel())
        if self._pipe is not None:
            n += int(self.lib.cama_pipeline_scratch_bytes(self._pipe["handle"]))
        return n

    def _scratch_need(self, N, F, rig):
        key = (N, F, rig.C, rig.H, rig.W, self.radius)
        memo = self.__dict__.setdefault("_need_memo", {})
        need = memo.get(key)
        if need is None:
            need = memo[key] = int(self.lib.cama_render_scratch_bytes(N, F, rig.C, rig.H, rig.W, self.radius))
        return need

    def _stage_poses(self, P, w2c):
        """Host float32 matrices (the np.linalg.inv result, cama/dataset.py:99) -> the pipeline's pose slot of the next
        launch (cama_pipeline_stage_poses): no torch tensor, no allocator traffic, a device address that is fixed per
        slot.  Returns (device pointer, F) or None
        when `w2c` is not a host float32 array."""
        if not (isinstance(w2c, np.ndarray) and w2c.dtype == np.float32):
            return None
        import ctypes
        a = np.ascontiguousarray(w2c).reshape(-1, 16)
        ptr = ctypes.c_void_p()
        _lib.check(self.lib.cama_pipeline_stage_poses(P["handle"], a.ctypes.data, a.shape[0], ctypes.byref(ptr)))
        return ptr.value, a.shape[0]

    def render_frames_pipelined(self, dmap, rig, w2c, src, out, cols=3, crop=None, segments=False):
        """Like render_frames, but through the library's two-stream pipeline (cama_pipeline_render): the binning half
        runs on one internal stream and the overlay half on another with double-buffered scratch, so call k+1's
        binning overlaps call k's overlay (HBM-bound) instead of queueing behind it.  `src` / `w2c` must be complete
        on the CURRENT stream at call time; `out` is complete only after join() (which makes the current stream wait
        for every overlay issued so far).  Host float32 `w2c` (what frame_poses returns) takes the staged-pose path:
        no per-call tensor."""
        torch = _torch()
        cropa = self._crop(crop)
        P = self._pipeline()
        with torch.cuda.device(self.device):
            staged = self._stage_poses(P, w2c)
            if staged is not None:
                T, (T_ptr, F) = None, staged
            else:
                T = w2c if (isinstance(w2c, torch.Tensor) and w2c.dtype == torch.float64 and w2c.is_cuda
                            and w2c.dim() == 2) else self._mats(w2c)
                T_ptr, F = T.data_ptr(), T.shape[0]
            assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
            assert tuple(src.shape) == (F, rig.C, rig.H, rig.W, 3)
            assert tuple(out.shape) == self.mosaic_shape(rig, F, cols) and out.is_contiguous()
            x, y, z, col, key, bnd, bflags = dmap.render_ptrs(cropa)
            if segments:
                # EXTENSION (no reference semantics): discs + one-pixel Bresenham segments between polyline neighbours
                if key is not None or not getattr(dmap, "has_links", False):
                    raise _lib.CamaHipError("segments need a map in draw order that carries its polyline links "
                                            "(ClipManager builds one; spatially sorted maps have no neighbours)")
                if self.alpha256 != 256:
                    raise _lib.CamaHipError("segments and translucent stamps are separate extensions")
                bflags |= _lib.BIN_SEGMENTS
                if segments == "wu":        # anti-aliased (Wu) segments, blended once by coverage: include/cama_hip.h
                    bflags |= _lib.BIN_SEGMENTS_WU
            # scratch: the pipeline's own (NULL, NULL) -- sized from what the launch's cull lets through when the map is
            # site-sized (the call then waits on the host for the pre-pass: include/cama_hip.h), else for the worst case
            _lib.call_retrying_oom(self.lib.cama_pipeline_render,
                P["handle"], x, y, z, dmap.is_f64, col, key, bnd, bflags, dmap.N, T_ptr, F, rig.c2cam.data_ptr(),
                rig.K.data_ptr(), rig.C, cropa.ctypes.data, rig.W, rig.H, src.data_ptr(), out.data_ptr(), cols,
                self.radius, self.halfwidth.ctypes.data, self.palette.ctypes.data, None, None, 0, self._stream())
            self._last_bin = (dmap.N, F, rig.C, rig.H, rig.W, bnd is not None, key is not None, None)
            # The internal streams are invisible to torch's caching allocator: what launch k reads / writes must stay
            # allocated until the library reports it complete (cama_pipeline_completed, a hipEventQuery over its ring
            # of per-launch events) -- however far ahead of the GPU the host is.
            seq = int(self.lib.cama_pipeline_issued(P["handle"]))
            P["keep"].append((seq, T, src, out, dmap, rig, ()))
            self._release_completed(P)
            return out

    # ------------------------------------------------------------------ one call per launch of a clip
    def clip_desc(self, dmap, rig, crop=None, cols=3, segments=False, raw=None):
        """The launch-invariant half of a clip's pipelined renders as a cama_clip (include/cama_hip.h): filled once, handed
        to render_clip_launch() for every launch.  raw = (H0, W0, cm_list) for raw sensor frames through the 3:5 kernel
        (returns None when the rig's maps are not of that form).  The returned object keeps what its pointers refer to."""
        cropa = self._crop(crop)
        x, y, z, col, key, bnd, bflags = dmap.render_ptrs(cropa)
        if segments:
            if key is not None or not getattr(dmap, "has_links", False):
                raise _lib.CamaHipError("segments need a map in draw order that carries its polyline links "
                                        "(ClipManager builds one; spatially sorted maps have no neighbours)")
            if self.alpha256 != 256:
                raise _lib.CamaHipError("segments and translucent stamps are separate extensions")
            bflags |= _lib.BIN_SEGMENTS | (_lib.BIN_SEGMENTS_WU if segments == "wu" else 0)
        d = _lib.Clip()
        d.x, d.y, d.z, d.colour_id, d.draw_key, d.block_bounds = x, y, z, col, key, bnd
        d.c2cam, d.K = rig.c2cam.data_ptr(), rig.K.data_ptr()
        d.N, d.xyz_is_f64, d.flags, d.C, d.W, d.H, d.cols, d.radius = dmap.N, dmap.is_f64, bflags, rig.C, rig.W, rig.H, cols, self.radius
        for k in range(6):
            d.crop[k] = float(cropa[k])
        for k, v in enumerate(self.halfwidth):
            d.halfwidth[k] = int(v)
        for k, v in enumerate(self.palette.reshape(-1)):
            d.palette_bgr[k] = int(v)
        keep = [dmap, rig]
        if raw is not None:
            H0, W0, cm_list = raw
            plan = self.rig_maps(cm_list)
            vrows = plan[8]
            if vrows is None:
                return None
            d.kind, d.H0, d.W0 = 1, int(H0), int(W0)
            d.vrows, d.band_rows, d.max_src_rows = vrows[0].data_ptr(), vrows[1].data_ptr(), int(vrows[2])
            keep.append(tuple(t for t in (vrows[0], vrows[1], plan[0], plan[1], plan[3], plan[5]) if t is not None))
        import ctypes
        d._keep = keep
        # (the address, not ctypes.byref(d): a byref object stored on the struct it points to is a reference cycle the
        # collector cannot see through -- CArgObject has no tp_traverse -- and would pin the map and the rig for ever)
        d._addr = ctypes.addressof(d)
        return d

    def render_clip_launch(self, desc, w2c_ptr, F, src_ptr, out_ptr, keep):
        """One pipelined launch of a clip in ONE library call (cama_pipeline_render_clip): w2c_ptr = HOST float32 [F,16].
        `keep` = the 6-tuple (None, src, out, dmap, rig, extra) of what the launch reads / writes (held until it completes)."""
        P = self._pipe or self._pipeline()
        torch = _torch()
        box = self.__dict__.get("_seq_box")
        if box is None:
            import ctypes
            box = self.__dict__["_seq_box"] = (ctypes.c_int64(0), ctypes.c_int64(0))
            box = box + (ctypes.byref(box[0]), ctypes.byref(box[1]))
            self.__dict__["_seq_box"] = box
        fn = self.lib.cama_pipeline_render_clip
        args = (P["handle"], desc._addr, w2c_ptr, F, src_ptr, out_ptr)
        if torch.cuda.current_device() == self.device.index:
            rc = fn(*args, torch.cuda.current_stream(self.device).cuda_stream, box[2], box[3])
            if rc == _lib.ENOMEM:
                self.pool.trim(0)                                   # idle pooled mosaics first: they are outside torch's cache
                torch.cuda.empty_cache()
                rc = fn(*args, torch.cuda.current_stream(self.device).cuda_stream, box[2], box[3])
        else:
            with torch.cuda.device(self.device):
                rc = fn(*args, self._stream(), box[2], box[3])
                if rc == _lib.ENOMEM:
                    self.pool.trim(0)
                    torch.cuda.empty_cache()
                    rc = fn(*args, self._stream(), box[2], box[3])
        if rc:
            _lib.check(rc)
        kp = P["keep"]
        kp.append((box[0].value,) + keep)
        done = box[1].value
        n = 0
        while n < len(kp) and kp[n][0] <= done:
            n += 1
        if n:
            del kp[:n]

    # ------------------------------------------------------------------ many scenes per launch
    def scene_batchable(self, items):
        """True when `items` = [(dmap, rig, w2c, src, out), ...] can go out as ONE multi-scene launch (cama_*_scenes):
        same frame count, rig geometry and vertex dtype everywhere, pre-resized frames, maps small enough to run without
        the block index (big site maps are launched per scene: there the launch is milliseconds, not its overhead)."""
        if len(items) < 2 or self.alpha256 != 256:
            return False
        d0, r0, w0, s0, _ = items[0]
        F = len(w0)
        if F == 0 or r0.W % 16:
            return False
        for dmap, rig, w2c, src, out in items:
            if (len(w2c) != F or (rig.C, rig.H, rig.W) != (r0.C, r0.H, r0.W) or dmap.is_f64 != d0.is_f64 or dmap.N == 0
                    or dmap.N >= BOUNDS_MIN_VERTS and getattr(dmap, "bounds", None) is not None
                    or tuple(src.shape) != (F, rig.C, rig.H, rig.W, 3) or not src.is_contiguous() or not out.is_contiguous()
                    or not (isinstance(w2c, np.ndarray) and w2c.dtype == np.float32)):
                return False
        return len(items) * F <= 65535 and len(items) <= MAX_SCENES_PER_LAUNCH

    def _scene_table(self, items):
        """(host uint64 [S,10], device twin) of cama_scene entries for `items`, cached on the pointers themselves."""
        torch = _torch()
        rows = []
        for dmap, rig, _, src, out in items:
            if dmap.sorted_soa is None:
                x, y, z = dmap.ptrs()
                col, key = dmap.colour.data_ptr(), 0
            else:
                x, y, z, col, key = dmap.render_ptrs()[:5]
            rows.append((x, y, z, col, key or 0, rig.c2cam.data_ptr(), rig.K.data_ptr(), src.data_ptr(), out.data_ptr(), dmap.N))
        host = np.asarray(rows, dtype=np.uint64)
        cache = self.__dict__.setdefault("_scene_tables", {})
        k = host.tobytes()
        hit = cache.get(k)
        if hit is None:
            if len(cache) >= 32:
                cache.pop(next(iter(cache)))
            hit = cache[k] = (host, torch.from_numpy(host.view(np.int64)).to(self.device))
        return hit

    def render_scenes(self, items, cols=3, crop=None, pipelined=True):
        """items = [(dmap, rig, w2c float32 host [F,4,4], src [F,C,H,W,3], out [F,2H,3W,3]), ...] -> ONE binning chain and
        ONE overlay launch for all of them (scene_batchable(items) must hold).  Bit-identical to rendering them one by
        one.  pipelined: through the two-stream pipeline (outputs complete after join()), else on the current stream."""
        torch = _torch()
        cropa = self._crop(crop)
        S = len(items)
        d0, r0, w0, _, _ = items[0]
        F = len(w0)
        Nmax = max(it[0].N for it in items)
        with torch.cuda.device(self.device):
            host, dev = self._scene_table(items)
            poses = np.ascontiguousarray(np.concatenate([np.asarray(it[2], np.float32).reshape(F, 16) for it in items]))
            if pipelined:
                P = self._pipeline()
                T_ptr = self._stage_poses(P, poses)[0]
                _lib.call_retrying_oom(self.lib.cama_pipeline_render_scenes,
                    P["handle"], host.ctypes.data, dev.data_ptr(), S, d0.is_f64, T_ptr, F, r0.C, cropa.ctypes.data, r0.W, r0.H,
                    cols, self.radius, self.halfwidth.ctypes.data, self.palette.ctypes.data, None, None, 0, self._stream())
                seq = int(self.lib.cama_pipeline_issued(P["handle"]))
                P["keep"].append((seq, None, dev, items[0][4], d0, r0,
                                  tuple(t for it in items for t in (it[3], it[4], it[0].soa, it[0].colour, it[0].sorted_soa,
                                                                     it[0].sorted_key, it[1].c2cam, it[1].K) if t is not None)
                                  + (host_keepalive(host),)))
                self._release_completed(P)
                return
            T = self._mats(poses.reshape(-1, 4, 4))
            scratch = self._scratch_buf(self._scratch_need(Nmax, S * F, r0))
            _lib.check(self.lib.cama_render_scenes(
                host.ctypes.data, dev.data_ptr(), S, d0.is_f64, T.data_ptr(), F, r0.C, cropa.ctypes.data, r0.W, r0.H, cols,
                self.radius, self.halfwidth.ctypes.data, self.palette.ctypes.data, scratch.data_ptr(), scratch.numel(),
                self._stream()))

    def _release_completed(self, P):
        done = int(self.lib.cama_pipeline_completed(P["handle"]))
        if done < 0:
            _lib.check(done)
        keep = P["keep"]
        n = 0
        while n < len(keep) and keep[n][0] <= done:
            n += 1
        if n:
            del keep[:n]

    def join(self):
        """Make the current stream wait for all pipelined overlays issued so far.  The references kept for them are
        handed back to torch's allocator with record_stream(current stream): a block is then reused only after work
        queued behind that wait, on whichever stream it was allocated."""
        if self._pipe is None:
            return
        torch = _torch()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.cama_pipeline_join(self._pipe["handle"], self._stream()))
            self._release_completed(self._pipe)
            cur = torch.cuda.current_stream(self.device)
            # the consumer reads its mosaics on THIS stream from here on: a base that is re-lent later under another stream must
            # be ordered behind it (MosaicPool._lend waits for an event of the stream noted here)
            for b in self.pool.bases:
                if b["stream"] is not None and not self.pool.idle(b):
                    b["stream"] = cur
            for _, T, src, out, dmap, rig, extra in self._pipe["keep"]:
                for t in (T, src, out, dmap.soa, dmap.colour, dmap.sorted_soa, dmap.sorted_key,
                          dmap.__dict__.get("_bounds"), rig.c2cam, rig.K) + tuple(extra):        # (never the lazy property)
                    if t is not None and hasattr(t, "record_stream"):
                        t.record_stream(cur)
            self._pipe["keep"].clear()

    def __del__(self):
        pipe = getattr(self, "_pipe", None)
        if pipe is not None:
            try:
                self.lib.cama_pipeline_destroy(pipe["handle"])
            except Exception:
                pass
            self._pipe = None

    def max_frames_per_call(self, dmap, rig, budget_bytes=None, resident_frames=True, src_bytes_per_frame=None,
                            pipelined=False):
        """Largest F whose per-call memory fits `budget_bytes` (and 32-bit stamp offsets).  Per frame: the worst-case
        stamp scratch (every vertex visible in every camera; the pipeline's slots when pipelined) and -- unless the source frames
        and the mosaic are already resident (`resident_frames`, the bench / streaming case) -- the decoded source batch
        and its mosaic slice, which a disk-backed source allocates per call.  Default budget: a quarter of the free HBM,
        at most 64 GB -- sized for 288 GB parts, so that 4*10^6-vertex site maps still render 40 frames per launch."""
        memo_key = (dmap.N, rig.C, rig.H, rig.W, budget_bytes, resident_frames, src_bytes_per_frame, pipelined)
        memo = self.__dict__.setdefault("_fpc_memo", {})
        if memo_key in memo:                    # (the free-memory probe is a driver call: once per shape is enough)
            return memo[memo_key]
        planned = (self.alpha256 == 256 and dmap.N >= BOUNDS_MIN_VERTS and getattr(dmap, "bounds", None) is not None
                   and dmap.site_sized(self.crop) and PLAN_SITE_MAPS)
        if planned and resident_frames:
            # site-sized maps are PLANNED (the pipeline sizes its own stamp scratch from what survives the launch's cull:
            # ~1 GB per 167 frames of the 10^6-vertex stress instead of 24 GB), so memory no longer bounds the launch;
            # what does is the overlay's bandwidth, which falls slowly with the bytes a launch walks (40 / 80 / 167 frames
            # of 1600x900: 0.784 / 0.779 / 0.767 of 8 TB/s on never-touched buffers) against one ~12 us kernel boundary per
            # launch.  engine.SITE_FRAMES_PER_LAUNCH overrides.
            memo[memo_key] = max(1, int(SITE_FRAMES_PER_LAUNCH))
            return memo[memo_key]
        if budget_bytes is None:
            free, _ = _torch().cuda.mem_get_info(self.device)
            budget_bytes = min(64 << 30, max(1 << 30, free // 4))
        per = max(1, self.lib.cama_render_scratch_bytes(dmap.N, 1, rig.C, rig.H, rig.W, self.radius))
        if pipelined:
            per *= PIPELINE_DEPTH
        if not resident_frames:
            src = src_bytes_per_frame if src_bytes_per_frame is not None else rig.C * rig.H * rig.W * 3
            per += int(src) + rig.C * rig.H * rig.W * 3
        f_budget = max(1, int(budget_bytes // per))
        f_off = max(1, int(((1 << 32) - 1) // max(1, rig.C * max(1, dmap.N) * 2)))
        cap = 65535 if resident_frames else 512            # a disk-backed batch is decoded and held as a whole
        memo[memo_key] = min(f_budget, f_off, cap)
        return memo[memo_key]

    def shrink_frames_per_call(self):
        """After an out-of-memory error: forget the memoised per-call sizes and the scratch buffers (so that a smaller
        call can allocate), wait for what is in flight."""
        torch = _torch()
        self.__dict__.pop("_fpc_memo", None)
        if self._pipe is not None:
            self.join()
            torch.cuda.current_stream(self.device).synchronize()
            self.lib.cama_pipeline_destroy(self._pipe["handle"])      # (gives its own scratch back)
            self._pipe = None
        self._scratch = None
        self._last_bin = None                   # (its statistics lived in the scratch that just went)
        self.pool.trim(0)                       # idle pooled mosaics go as well
        torch.cuda.empty_cache()

    def stamp_points(self, image, vu, colour_id, link=None, wu=False):
        """In-place render_maps on one device image [H,W,3]: points (n,2) (v,u) float64 in draw order.  EXTENSION:
        `link` (n,) bool -- point k with link[k] is also joined to point k - 1 by a one-pixel Bresenham segment
        (cama_stamp_polylines; no reference semantics), or -- wu=True -- by an anti-aliased Wu line whose coverages are
        blended once per pixel (cama_stamp_polylines_wu)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            assert image.is_cuda and image.dtype == torch.uint8 and image.is_contiguous() and image.shape[2] == 3
            H, W = int(image.shape[0]), int(image.shape[1])
            p = torch.from_numpy(np.ascontiguousarray(np.asarray(vu, np.float64).reshape(-1, 2))).to(self.device)
            col = torch.from_numpy(np.ascontiguousarray(np.asarray(colour_id, np.uint8))).to(self.device)
            n = p.shape[0]
            lk = None
            if link is not None:
                lk = torch.from_numpy(np.ascontiguousarray(np.asarray(link, np.uint8))).to(self.device)
                assert lk.shape[0] == n
            need = self.lib.cama_stamp_scratch_bytes(H, W)
            scratch = self._scratch_buf(need)
            fn = self.lib.cama_stamp_polylines_wu if wu else self.lib.cama_stamp_polylines
            _lib.check(fn(p.data_ptr(), col.data_ptr(), None if lk is None else lk.data_ptr(), n,
                          image.data_ptr(), H, W, self.radius, self.halfwidth.ctypes.data,
                          self.palette.ctypes.data, scratch.data_ptr(), scratch.numel(), self._stream()))
            return image
