"""Camera-frame ingest for the overlay stage.

The reference reads one JPEG per camera per frame with cv2.imread and resamples it with
cv2.initUndistortRectifyMap + cv2.remap (cama/reproject.py:228-244).  OpenCV is installed on neither box
(SURVEY.md section 8c), so decode uses cv2 when importable, else Pillow (both wrap libjpeg; parity with
cv2.imread is unpinned), and `.npy` raw BGR frames are accepted next to / instead of the .jpg.

A FrameSource hands the overlay kernel a device tensor [F,C,H,W,3] uint8 BGR for a list of frame indices:
  * ClipFrameSource  -- reads <clip>/<camera>/<timestamp>.jpg|.npy (the demo's layout)
  * DeviceFrameSource -- frames already resident in HBM (bench / streaming pipelines)
"""
import os

import numpy as np


def read_rgb_or_bgr(path):
    """(array (H,W,3) uint8, is_rgb): decode without the host-side channel flip (Pillow decodes to RGB; the flip to the
    reference's BGR order is then done on the device).  cv2 / .npy sources are BGR already."""
    stem = os.path.splitext(path)[0]
    if not os.path.exists(path) and os.path.exists(stem + ".npy"):
        return np.load(stem + ".npy"), False
    if path.endswith(".npy"):
        return np.load(path), False
    try:
        import cv2
        img = cv2.imread(path)
        if img is None:
            raise FileNotFoundError(path)
        return img, False
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            return np.array(im.convert("RGB")), True        # own, writable buffer (copied inside the worker thread)


def _read_jpeg_bytes_or_array(path):
    """(bytes, False) for an existing JPEG file (to be decoded on the device), else what read_rgb_or_bgr returns."""
    if path.lower().endswith((".jpg", ".jpeg")) and os.path.exists(path):
        with open(path, "rb") as f:
            return f.read(), False
    return read_rgb_or_bgr(path)


def _read_into(blob, path):
    """Fill an ArenaBlob (pinned memory) with the file's bytes; (blob, False) like the other fetchers."""
    view = memoryview(blob.view())
    with open(path, "rb", buffering=0) as f:
        got = 0
        while got < blob.n:
            k = f.readinto(view[got:])
            if not k:
                break
            got += k
        # the slice was sized from a directory scan: a file that changed since then is read again, whole, the plain way
        if got == blob.n and not f.read(1):
            return blob, False
    return _read_jpeg_bytes_or_array(path)


def _read_batch_native(blobs, paths, C, threads):
    """All files of a batch into their arena slices through cama_read_files (the GIL is released for the whole call);
    returns [(frame k, camera c, blob or fallback array, is_rgb)] in (frame, camera) order."""
    import ctypes
    from . import _lib
    n = len(paths)
    L = _lib.lib()
    base = blobs[0].arena.np.ctypes.data
    P = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    D = (ctypes.c_void_p * n)(*[base + b.off for b in blobs])
    sizes = np.fromiter((b.n for b in blobs), dtype=np.uint64, count=n)
    status = np.ones(n, np.int32)
    _lib.check(L.cama_read_files(P, D, sizes.ctypes.data, n, int(threads), status.ctypes.data))
    out = []
    for k in range(n):
        if status[k] == 0:
            out.append((k // C, k % C, blobs[k], False))
        else:                                              # changed since the directory scan, unreadable, ...
            out.append((k // C, k % C) + tuple(_read_jpeg_bytes_or_array(paths[k])))
    return out


def _read_frame(items):
    """The camera files of one frame, one after the other (one pool task per frame instead of one per file: the pool's
    per-task cost on the submitting thread was the largest item of the main.py loop's host profile)."""
    return [_read_into(blob, path) for blob, path in items]


class _Part:
    """Camera c's share of a _read_frame task, with the two Future methods the consumers use."""
    __slots__ = ("fut", "c")

    def __init__(self, fut, c):
        self.fut, self.c = fut, c

    def result(self):
        return self.fut.result()[self.c]

    def cancel(self):
        return self.fut.cancel()


def _decode_bytes(data):
    import io
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:
        return np.array(im.convert("RGB"))


def read_bgr(path):
    """(H,W,3) uint8 BGR from a .jpg/.png path; falls back to the `.npy` twin of the same stem."""
    stem = os.path.splitext(path)[0]
    if not os.path.exists(path) and os.path.exists(stem + ".npy"):
        return np.ascontiguousarray(np.load(stem + ".npy"))
    if path.endswith(".npy"):
        return np.ascontiguousarray(np.load(path))
    try:
        import cv2
        img = cv2.imread(path)
        if img is None:
            raise FileNotFoundError(path)
        return img
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])


def undistort_rectify_map(K_origin, dist, K_new, W, H):
    """float32 (mapx, mapy), each (H,W): source pixel of every destination pixel -- what
    cv2.initUndistortRectifyMap(K_origin, dist, None, K_new, (W,H), CV_32FC1) computes (reproject.py:238), following
    OpenCV's scalar loop operation for operation (imgproc undistort: closed-form 3x3 inverse of K_new, row-incremental
    _x/_y/_w, w = 1./_w, rational + tangential + thin-prism model, u = fx*xd + u0 with NO skew term), vectorised:
    the row-incremental sums are np.add.accumulate along the row, which adds in the same order.  Bit-identical to the
    checker's C restatement (oracle_undistort_map).  For zero distortion and K_new = diag(sx, sy, 1) K_origin this is
    src = dst / scale.  The tilted-sensor coefficients (dist[12:14]) are not supported and raise."""
    K0 = np.asarray(K_origin, np.float64).reshape(3, 3)
    S = [float(v) for v in np.asarray(K_new, np.float64).reshape(9)]
    k = np.zeros(14)
    d = np.asarray([] if dist is None else dist, np.float64).reshape(-1)
    k[:min(14, d.size)] = d[:14]
    if k[12] != 0.0 or k[13] != 0.0:
        raise NotImplementedError("tilted-sensor distortion coefficients (tauX, tauY) are not supported")
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = (float(v) for v in k[:12])
    # cv::invert, 3x3 closed form: inverse = adjugate * (1/det)
    det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6])
    if det == 0.0:
        raise ValueError("singular new camera matrix")
    dd = 1.0 / det
    ir = [(S[4] * S[8] - S[5] * S[7]) * dd, (S[2] * S[7] - S[1] * S[8]) * dd, (S[1] * S[5] - S[2] * S[4]) * dd,
          (S[5] * S[6] - S[3] * S[8]) * dd, (S[0] * S[8] - S[2] * S[6]) * dd, (S[2] * S[3] - S[0] * S[5]) * dd,
          (S[3] * S[7] - S[4] * S[6]) * dd, (S[1] * S[6] - S[0] * S[7]) * dd, (S[0] * S[4] - S[1] * S[3]) * dd]
    i = np.arange(H, dtype=np.float64)

    def along_rows(start, step):                     # _v = start_i, then `_v += step` per column
        a = np.full((H, W), step, np.float64)
        a[:, 0] = start
        return np.add.accumulate(a, axis=1)
    _x = along_rows(i * ir[1] + ir[2], ir[0])
    _y = along_rows(i * ir[4] + ir[5], ir[3])
    _w = along_rows(i * ir[7] + ir[8], ir[6])
    w = 1.0 / _w
    x, y = _x * w, _y * w
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2) + s1 * r2 + s2 * r2 * r2
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy + s3 * r2 + s4 * r2 * r2
    u = float(K0[0, 0]) * 1.0 * xd + float(K0[0, 2])
    v = float(K0[1, 1]) * 1.0 * yd + float(K0[1, 2])
    return np.ascontiguousarray(u.astype(np.float32)), np.ascontiguousarray(v.astype(np.float32))


def undistort_rectify_map_separable(K_origin, dist, K_new, W, H):
    """(u[W], v[H]) float32 with undistort_rectify_map(...) == (broadcast of u over the rows, broadcast of v over the
    columns) BIT FOR BIT, or None when the map is not of that form.

    It is of that form when every distortion coefficient is zero (the nuScenes cameras, and what the reference's
    CameraManager undistorts TO) and the inverse of K_new has no skew / projective terms (ir[1] = ir[3] = ir[6] = ir[7] = 0):
    then in OpenCV's loop _x depends on the column only (start ir[2], += ir[0]), _y on the row only (its per-column
    increment ir[3] is an exact + 0.0), _w is the constant ir[8], kr = 1 / 1, and the tangential / prism terms add exact
    zeros -- so one row and one column carry the whole map.  O(W + H) instead of O(W H): 80 ms per six-camera rig at 960x540
    on the host, paid by every new clip's first frame (profiles/r05_cold_sweep.txt), becomes 0.2 ms."""
    d = np.asarray([] if dist is None else dist, np.float64).reshape(-1)
    if d.size and np.any(d[:14] != 0.0):
        return None
    S = [float(v) for v in np.asarray(K_new, np.float64).reshape(9)]
    det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6])
    if det == 0.0:
        raise ValueError("singular new camera matrix")
    dd = 1.0 / det
    ir1, ir3 = (S[2] * S[7] - S[1] * S[8]) * dd, (S[5] * S[6] - S[3] * S[8]) * dd
    ir6, ir7 = (S[3] * S[7] - S[4] * S[6]) * dd, (S[1] * S[6] - S[0] * S[7]) * dd
    if ir1 != 0.0 or ir3 != 0.0 or ir6 != 0.0 or ir7 != 0.0:
        return None
    u, _ = undistort_rectify_map(K_origin, dist, K_new, W, 1)        # row 0: i = 0
    _, v = undistort_rectify_map(K_origin, dist, K_new, 1, H)        # column 0: j = 0
    return np.ascontiguousarray(u[0]), np.ascontiguousarray(v[:, 0])


def camera_maps_compact(cm):
    """(mapx, mapy, separable): separable = 1 -> mapx is a W-vector and mapy an H-vector (zero distortion: see
    undistort_rectify_map_separable), else both are (H, W) planes.  Built once and cached on the CameraManager."""
    hit = getattr(cm, "_resample_maps_compact", None)
    if hit is not None:
        return hit
    full = getattr(cm, "_resample_maps", None)
    if full is None:
        d = cm.d_origin if cm.d == [] else cm.d
        sep = undistort_rectify_map_separable(cm.K_origin, [] if d is None else d, cm.K, cm.width, cm.height)
        if sep is not None:
            cm._resample_maps_compact = (sep[0], sep[1], 1)
            return cm._resample_maps_compact
        full = camera_maps(cm)
    mx, my = full
    if bool((mx == mx[0:1, :]).all() and (my == my[:, 0:1]).all()):
        hit = (np.ascontiguousarray(mx[0, :]), np.ascontiguousarray(my[:, 0]), 1)
    else:
        hit = (mx, my, 0)
    cm._resample_maps_compact = hit
    return hit


def camera_maps(cm):
    """(mapx, mapy) of a CameraManager, built once and cached on it."""
    maps = getattr(cm, "_resample_maps", None)
    if maps is None:
        d = cm.d_origin if cm.d == [] else cm.d
        maps = undistort_rectify_map(cm.K_origin, [] if d is None else d, cm.K, cm.width, cm.height)
        cm._resample_maps = maps
    return maps


def resample_host_image(cm, image):
    """Undistort + resize one host image to the CameraManager's output size on the device."""
    import torch
    from . import runtime
    eng = runtime.engine()
    src = torch.from_numpy(np.ascontiguousarray(image)).to(eng.device)
    return eng.resample(cm, src).cpu().numpy()


class DeviceFrameSource:
    """Frames resident in HBM: tensor [F_total, C, H, W, 3] uint8; index i = sync image index i."""

    def __init__(self, frames, index_offset=0):
        self.frames = frames
        self.index_offset = index_offset
        self._placed = False

    def place_for(self, eng, rig, out, image_indices):
        """Once per source: move the resident frames into the allocation from which the overlay into `out` (the mosaic of
        the frames `image_indices`, one contiguous run) is fastest (Engine.place_frames: the source's placement is worth
        2-4 % once the mosaic's is good; buffers of 512 MiB .. 8 GiB, CAMA_AUDITION=0 switches it off).  The caller's own
        tensor is left alone -- the source simply reads the placed copy from now on."""
        if self._placed:
            return
        import os
        K = int(os.environ.get("CAMA_AUDITION", "16")) // 2
        idx = [int(i) - self.index_offset for i in image_indices]
        if K <= 1 or not idx or idx != list(range(idx[0], idx[0] + len(idx))) or len(idx) != int(out.shape[0]):
            return                                  # (not this time: a later whole-run render may still place the frames)
        self._placed = True
        self.frames = eng.place_frames(rig, self.frames, out, first=idx[0], candidates=K)

    def batch(self, image_indices):
        idx = [i - self.index_offset for i in image_indices]
        if len(idx) and idx == list(range(idx[0], idx[0] + len(idx))):
            return self.frames[idx[0]:idx[0] + len(idx)]          # contiguous view, no copy
        import torch
        return self.frames[torch.as_tensor(idx, device=self.frames.device)].contiguous()


class ClipFrameSource:
    """Decode + upload the demo's per-camera frame files (<clip>/<camera>/<timestamp>.jpg|.npy).

    Decode is the real bottleneck of the demo once the reprojection runs on the GPU (examples/demo_synthetic.py:
    ~12 ms per 1600x900 JPEG on one host core, 96 % of a frame).  libjpeg releases the GIL, so the cameras of a frame
    are decoded by a thread pool, and the next frame's files are prefetched while the current frame is consumed.
    Raw frames go to the device as they are; undistort + resize happens inside the overlay kernel
    (cama_overlay_frames_raw) when the output size differs from the sensor size."""

    def __init__(self, cm_list, device, workers=None, prefetch=3, decoder=None):
        self.cm_list = cm_list
        self.device = device
        self.fused = any(cm.needs_resample() for cm in cm_list)     # hand RAW frames to the fused overlay
        # "device": JPEG files are read as bytes and decoded on the GPU (cama_amd.jpeg, byte-identical to libjpeg-turbo);
        # "host": decoded by the worker threads.  CAMA_JPEG_DECODER overrides the default.
        self.decoder = decoder or os.environ.get("CAMA_JPEG_DECODER", "device" if device is not None else "host")
        assert self.decoder in ("device", "host")
        self._jpeg = None
        self._ahead = {}                                              # tuple(image indices) -> PendingDecode
        self._decode_depth = 2                                        # batches decoded ahead of the consumer
        self._pool = None
        # host decode: 12 threads peak (~500 images/s; the GIL beyond that).  Device decode: the workers only read
        # files, a few are enough (more just burn the container's CPU quota)
        self._workers = workers or (4 if self.decoder == "device" else min(12, (os.cpu_count() or 4)))
        # planned passes (plan()): a pump thread reads, parses and submits the device decode of the next batches while the
        # consumer renders / encodes the current one
        self._plan = None
        self._pump_depth = 3                                             # decoded or decoding batches ahead of the consumer
        self._pump_groups = 1                                            # decode groups per pumped batch (None: the decoder's own rule)
        self.use_pump = True                                             # (False: every batch on the unplanned path)
        self._batch_reads = {}                                           # batch key -> future of its file reads
        self._native_threads = 8                                         # native reader threads per batch
        self._prefetch = prefetch
        self._pending = {}                                            # image index -> list of futures
        self._dir_sizes = {}                                          # directory -> {file name: bytes} (one scan each)
        # decoded frames stay in HBM for the later passes over the clip (main.py renders every clip twice: the CAMA pass and
        # the nuScenes pass read the SAME camera files, main.py:57,66): batch key -> (tensor [F,C,H0,W0,3], per-frame file
        # signatures).  A 40-frame scene is 1 GB of 288; CAMA_FRAME_CACHE_BYTES (default 8 GiB, 0 = off) bounds it, oldest
        # batch first.  A frame is served from the cache only while the size and mtime of its six files are what they were
        # when it was decoded (the reference re-reads the files on every pass).
        import collections
        self._cache = collections.OrderedDict()
        self._cache_of = {}                                           # image index -> (batch key, row)
        self._cache_bytes = 0
        self._cache_cap = int(float(os.environ.get("CAMA_FRAME_CACHE_BYTES", str(8 << 30)))) if device is not None else 0
        self.cache_stats = {"hits": 0, "misses": 0, "stale": 0, "stored_batches": 0, "evicted_batches": 0}

    def _shared(self):
        """The process-wide Engine when it drives this source's GPU (ingest state is shared through it), else None."""
        if self.device is None:
            return None
        try:
            from . import runtime
            eng = runtime.engine()
        except Exception:
            return None
        import torch
        return eng if torch.device(self.device) == eng.device else None

    def _executor(self):
        if self._pool is None:
            eng = self._shared() if self.decoder == "device" else None
            if eng is not None:
                self._pool, self._own_pool = eng.reader_pool(self._workers), False
            else:
                from concurrent.futures import ThreadPoolExecutor
                self._pool, self._own_pool = ThreadPoolExecutor(max_workers=self._workers, thread_name_prefix="cama-decode"), True
        return self._pool

    def _decoder(self):
        if self._jpeg is None:
            eng = self._shared()
            if eng is not None:
                self._jpeg = eng.jpeg_decoder()       # one decoder per GPU: lanes, pinned arenas and scratch outlive the clip
            else:
                from .jpeg import DeviceJpegDecoder
                self._jpeg = DeviceJpegDecoder(self.device)
        return self._jpeg

    def _submit(self, idx):
        """Start reading the C camera files of sync index `idx`.  Device decoder: JPEG files are read by the workers
        STRAIGHT INTO pinned memory (a slice of the decoder's current arena, sized from os.stat), so the compressed
        bytes are copied once (page cache -> pinned) and uploaded from where they are; anything else (.npy twins,
        host decoder) goes through the old fetchers."""
        if idx in self._pending:
            return self._pending[idx]
        ex = self._executor()
        paths = [cm.get_image_path(idx, True) for cm in self.cm_list]
        if self.decoder != "device":
            self._pending[idx] = [ex.submit(read_rgb_or_bgr, p) for p in paths]
            return self._pending[idx]
        sizes = [self._file_size(p) if p.lower().endswith((".jpg", ".jpeg")) else -1 for p in paths]
        futs = []
        if min(sizes) > 0:
            need = sum(sz + 16 for sz in sizes)
            a = getattr(self, "_arena", None)
            if a is None or a.size - a.used < need:
                a = self._arena = self._decoder().arena(max(need, 96 << 20))
            fut = ex.submit(_read_frame, [(a.take(sz), p) for p, sz in zip(paths, sizes)])
            futs = [_Part(fut, c) for c in range(len(paths))]
        else:
            futs = [ex.submit(_read_jpeg_bytes_or_array, p) for p in paths]
        self._pending[idx] = futs
        return futs

    def _file_size(self, path):
        """Size of a camera file from one scan of its directory (an os.stat per file was 8.5 us x 6 cameras x every frame
        on the submitting thread: 12 % of the main.py loop); -1 when it cannot be had.  _read_into re-checks the length."""
        d, name = os.path.split(path)
        tab = self._dir_sizes.get(d)
        if tab is None:
            tab = self._dir_sizes[d] = {}
            try:
                with os.scandir(d) as it:
                    for e in it:
                        try:
                            tab[e.name] = e.stat().st_size
                        except OSError:
                            pass
            except OSError:
                pass
        sz = tab.get(name)
        if sz is None:
            try:
                sz = tab[name] = os.stat(path).st_size
            except OSError:
                sz = -1
        return sz

    def _n_frames(self):
        return len(self.cm_list[0].dr.attribute["sync"][self.cm_list[0].camera_name])

    def _collect(self, image_indices):
        """[(frame k, camera c, array, is_rgb)] of the requested frames (decoded in parallel, next frames prefetched)."""
        idx = [int(i) for i in image_indices]
        for i in idx:
            self._submit(i)
        for ahead in range(1, int(self._prefetch) + 1):              # keep the decode workers busy
            if idx and idx[-1] + ahead < self._n_frames():
                self._submit(idx[-1] + ahead)
        out = []
        for k, i in enumerate(idx):
            for c, fut in enumerate(self._pending.pop(i)):
                arr, is_rgb = fut.result()
                out.append((k, c, arr, is_rgb))
        for stale in [k for k in self._pending if k < idx[0]]:       # never consumed (skipped frames)
            for fut in self._pending.pop(stale):
                fut.cancel()
        return out

    def _decoded(self, image_indices):
        """host uint8 BGR array [F,C,H0,W0,3] of the requested frames."""
        items = self._collect(image_indices)
        F = len(list(image_indices))
        from .jpeg import as_bytes, is_blob
        items = [(k, c, _decode_bytes(as_bytes(arr)), True) if is_blob(arr) else (k, c, arr, is_rgb)
                 for k, c, arr, is_rgb in items]
        first = items[0][2]
        host = np.empty((F, len(self.cm_list)) + first.shape, np.uint8)
        for k, c, arr, is_rgb in items:
            host[k, c] = arr[:, :, ::-1] if is_rgb else arr
        return host

    # ------------------------------------------------------------------ planned passes: the decode pump
    def plan(self, batches):
        """Announce the batches raw_batch() will be asked for, in order (ClipManager knows every frame of a pass up front).
        A pump thread then walks the plan ahead of the consumer: file reads for the next batches (straight into pinned
        memory), header parsing, upload and the device decode launches all happen there, so the consumer's raw_batch()
        finds its batch decoded or decoding instead of doing that work between two renders (profiles/r03_demo_loop_*:
        45 % of the main.py loop was the consumer waiting for file reads, 20 % submitting decodes).  Device decoder only;
        a request that leaves the announced order cancels the plan and takes the unplanned path."""
        import threading
        self.cancel_plan()
        keys = [tuple(int(i) for i in b) for b in batches if len(b)]
        # batches whose frames are all still cached from an earlier pass never reach the pump (raw_batch serves them first)
        keys = [k for k in keys if self._cache_lookup(list(k), count=False) is None]
        if self.decoder != "device" or not keys or not self.use_pump:
            return
        P = {"keys": keys, "index": {k: j for j, k in enumerate(keys)}, "ready": {}, "consumed": 0,
             "cv": threading.Condition(), "stop": False, "error": None}
        # the thread holds only a WEAK reference to this source (and wakes up now and then to look at it): a consumer that
        # abandons a pass -- break, exception, dropping the ClipManager -- must not leave a parked thread that keeps the
        # source, its decoded batches (~0.4 GB of HBM each), pinned arenas and decoder lanes alive for ever
        import weakref
        P["thread"] = threading.Thread(target=_pump_entry, args=(weakref.ref(self), P), daemon=True, name="cama-decode-pump")
        self._plan = P
        P["thread"].start()

    def cancel_plan(self):
        P, self._plan = self._plan, None
        if P is None:
            return
        import threading
        with P["cv"]:
            P["stop"] = True
            P["cv"].notify_all()
        if P["thread"] is not threading.current_thread():      # (the pump itself may drop the last reference to the source)
            P["thread"].join()
        for pend in P["ready"].values():                 # release their lanes
            if hasattr(pend, "result"):
                try:
                    pend.result()
                except Exception:
                    pass
        P["ready"].clear()
        for futs in self._pending.values():
            for f in futs:
                f.cancel()
        self._pending.clear()
        for f in self._batch_reads.values():               # (a started read finishes into its arena slice: harmless)
            f.cancel()
        self._batch_reads.clear()

    def _pump_step(self, P, j, key):
        """One batch of the plan, on the pump thread: reads (this batch and the next two), then parse + upload + decode."""
        import torch
        from .jpeg import is_blob
        with torch.cuda.device(self.device):
            keys = P["keys"]
            for ahead in keys[j:j + 3]:            # file reads: this batch and the next two
                self._submit_batch(ahead)
            items = self._batch_reads.pop(key).result()
            if all(is_blob(arr) for _, _, arr, _ in items):
                # on a stream of the pump's own: the decoder orders its lanes behind the CALLER's current stream, and the
                # device's default stream is where the consumer thread queues its renders and the 150 MB downloads of
                # its mosaics -- every decode used to start behind whatever of those was queued (round 4: the loop ran
                # at decode + download, not at the larger of the two)
                if getattr(self, "_pump_stream", None) is None:
                    eng = self._shared()
                    # (the engine's: a stream per clip would be a pool of its own in torch's caching allocator)
                    self._pump_stream = eng.pump_stream() if eng is not None else torch.cuda.Stream(device=self.device)
                with torch.cuda.stream(self._pump_stream):
                    # one group per batch: the pump keeps `_pump_depth` batches in flight, that is the concurrency
                    return self._decoder().decode_async([arr for _, _, arr, _ in items], bgr=True, groups=self._pump_groups)
            return items                           # .npy twins / mixed sources: the consumer finishes them

    def close(self):
        """Stop the decode pump and the reader pool; safe to call more than once."""
        try:
            self.cancel_plan()
        finally:
            pool, self._pool = self._pool, None
            if pool is not None and getattr(self, "_own_pool", True):        # (the engine's shared pool stays)
                pool.shutdown(wait=False, cancel_futures=True)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _submit_batch(self, key):
        """Start reading all camera files of the frames `key` (a tuple of image indices): ONE pool task that hands the
        whole batch to cama_read_files -- native threads, the interpreter lock released for the duration -- reading
        straight into slices of the pinned arena.  (One Python task per frame cost ~25 us of interpreter time per file:
        with 1440 files per 240-frame pass that alone serialised ~40 ms per pass behind the GIL, next to the pump's and
        the consumer's own Python.)  Files that are not JPEGs of a known size, or that changed since the directory scan,
        go through the ordinary readers."""
        if key in self._batch_reads:
            return
        paths = [cm.get_image_path(i, True) for i in key for cm in self.cm_list]
        sizes = [self._file_size(p) if p.lower().endswith((".jpg", ".jpeg")) else -1 for p in paths]
        C = len(self.cm_list)
        if min(sizes) <= 0:
            self._batch_reads[key] = self._executor().submit(
                lambda: [(k // C, k % C) + tuple(_read_jpeg_bytes_or_array(p)) for k, p in enumerate(paths)])
            return
        need = sum(sz + 16 for sz in sizes)
        a = getattr(self, "_arena", None)
        if a is None or a.size - a.used < need:
            a = self._arena = self._decoder().arena(max(need, 128 << 20))
        blobs = [a.take(sz) for sz in sizes]
        self._batch_reads[key] = self._executor().submit(_read_batch_native, blobs, paths, C, self._native_threads)

    def _planned(self, key):
        """The pump's result for `key` if it is the plan's next batch, else None (the plan is cancelled when the request
        leaves the announced order)."""
        P = self._plan
        if P is None:
            return None
        if P["index"].get(key) != P["consumed"]:
            self.cancel_plan()
            return None
        with P["cv"]:
            while key not in P["ready"] and P["error"] is None:
                P["cv"].wait()
            if P["error"] is not None:
                err = P["error"]
            else:
                err = None
                pend = P["ready"].pop(key)
                P["consumed"] += 1
                P["cv"].notify_all()
        if err is not None:
            self.cancel_plan()
            raise err
        if P["consumed"] == len(P["keys"]):                # the pass is over: the pump has returned
            self._plan = None
        return pend

    # ------------------------------------------------------------------ decoded frames kept for the clip's later passes
    def _frame_signature(self, idx):
        """(size, mtime_ns) of the C camera files of sync index `idx`, or None when one cannot be stat'ed."""
        sig = []
        for cm in self.cm_list:
            try:
                st = os.stat(cm.get_image_path(idx, True))
            except OSError:
                return None
            sig.append((st.st_size, st.st_mtime_ns))
        return tuple(sig)

    def _cache_drop(self, key):
        t, _ = self._cache.pop(key)
        self._cache_bytes -= int(t.numel())
        for i in key:
            if self._cache_of.get(i, (None,))[0] == key:
                del self._cache_of[i]

    def _cache_store(self, image_indices, tensor):
        nbytes = int(tensor.numel())
        if self._cache_cap <= 0 or nbytes > self._cache_cap:
            return
        key = tuple(image_indices)
        sigs = [self._frame_signature(i) for i in key]
        if any(s is None for s in sigs):
            return
        if key in self._cache:
            self._cache_drop(key)
        while self._cache and self._cache_bytes + nbytes > self._cache_cap:
            self._cache_drop(next(iter(self._cache)))
            self.cache_stats["evicted_batches"] += 1
        self._cache[key] = (tensor, sigs)
        self._cache_bytes += nbytes
        for row, i in enumerate(key):
            self._cache_of[i] = (key, row)
        self.cache_stats["stored_batches"] += 1

    def _cache_lookup(self, image_indices, count=True):
        """The decoded frames `image_indices` from HBM when every one of them is cached and its files are unchanged: a view
        when they are consecutive rows of one cached batch, else one gather; None otherwise."""
        if not self._cache:
            return None
        where = [self._cache_of.get(i) for i in image_indices]
        if any(w is None for w in where):
            return None
        for i, (key, row) in zip(image_indices, where):
            if self._frame_signature(i) != self._cache[key][1][row]:
                self._cache_drop(key)                                 # a file changed: that batch is decoded again
                self.cache_stats["stale"] += 1
                return None
        if count:
            self.cache_stats["hits"] += 1
        import torch
        k0, r0 = where[0]
        if all(k == k0 and r == r0 + j for j, (k, r) in enumerate(where)):
            self._cache.move_to_end(k0)
            return self._cache[k0][0][r0:r0 + len(where)]
        return torch.stack([self._cache[k][0][r] for k, r in where])

    def raw_batch(self, image_indices):
        """device uint8 BGR tensor [F,C,H0,W0,3]: every image goes to its slot as decoded (no host-side stacking or
        channel flip: both hold the GIL the decode threads need); RGB-decoded frames are flipped on the device."""
        image_indices = [int(i) for i in image_indices]
        hit = self._cache_lookup(image_indices)
        if hit is not None:
            return hit
        if self._cache_cap > 0:
            self.cache_stats["misses"] += 1
        out = self._raw_batch_decode(image_indices)
        self._cache_store(image_indices, out)
        return out

    def _raw_batch_decode(self, image_indices):
        import torch
        F = len(image_indices)
        planned = self._planned(tuple(image_indices))
        if planned is not None:
            if hasattr(planned, "result"):
                flat = planned.result()
                return flat.view((F, len(self.cm_list)) + tuple(flat.shape[1:]))
            return self._upload_items(planned, F)
        ahead = self._ahead.pop(tuple(image_indices), None) if self._ahead else None
        if ahead is not None:                                        # decoded while the previous batch was consumed
            flat = ahead.result()
            self._decode_ahead(image_indices)
            return flat.view((F, len(self.cm_list)) + tuple(flat.shape[1:]))
        from .jpeg import as_bytes, is_blob
        items = self._collect(image_indices)
        if all(is_blob(arr) for _, _, arr, _ in items):
            # compressed bytes straight to the device decoder: one upload + one decode for the whole batch, BGR out
            flat = self._device_decode(image_indices, items)
            return flat.view((F, len(self.cm_list)) + tuple(flat.shape[1:]))
        return self._upload_items(items, F)

    def _upload_items(self, items, F):
        """Host-decoded / mixed items [(frame k, camera c, array or JPEG bytes, is_rgb)] -> device BGR tensor."""
        import torch
        from .jpeg import as_bytes, is_blob
        items = [(k, c, _decode_bytes(as_bytes(arr)) if is_blob(arr) else arr,
                  True if is_blob(arr) else is_rgb) for k, c, arr, is_rgb in items]
        shape = items[0][2].shape
        dev = torch.empty((F, len(self.cm_list)) + tuple(shape), dtype=torch.uint8, device=self.device)
        rgb = torch.zeros((F, len(self.cm_list)), dtype=torch.bool)
        for k, c, arr, is_rgb in items:
            dev[k, c].copy_(torch.from_numpy(arr), non_blocking=False)
            rgb[k, c] = is_rgb
        if bool(rgb.all()):
            dev = dev.flip(-1).contiguous()
        elif bool(rgb.any()):
            m = rgb.to(self.device)
            dev[m] = dev[m].flip(-1)
        return dev

    def _device_decode(self, image_indices, items):
        flat = self._decoder().decode([arr for _, _, arr, _ in items], bgr=True)
        self._decode_ahead(image_indices)
        return flat

    def _decode_ahead(self, image_indices):
        """Start decoding the batch that follows `image_indices` (same length, consecutive frames) on the decoder's
        side streams: the decode kernels are latency-bound and leave most of the GPU idle, so the next frame's decode
        runs under the current frame's host work and overlay."""
        step = len(image_indices)
        if image_indices != list(range(image_indices[0], image_indices[0] + step)):
            return
        wanted = []
        for d in range(1, self._decode_depth + 1):                       # the next `depth` batches
            nxt = [i + d * step for i in image_indices]
            if nxt[-1] >= self._n_frames():
                break
            wanted.append(tuple(nxt))
        for stale in [k for k in self._ahead if k not in wanted]:
            self._ahead.pop(stale).result()                              # release its lanes
        for key in wanted:
            if key in self._ahead:
                continue
            from .jpeg import is_blob
            items = self._collect(list(key))
            if all(is_blob(arr) for _, _, arr, _ in items):
                self._ahead[key] = self._decoder().decode_async([arr for _, _, arr, _ in items], bgr=True)
            else:                                                        # mixed sources: leave them to the normal path
                self._requeue(list(key), items)
                break

    def _requeue(self, idx, items):
        from concurrent.futures import Future
        for i in idx:
            futs = []
            for k, c, arr, is_rgb in items:
                if idx[k] == i:
                    f = Future()
                    f.set_result((arr, is_rgb))
                    futs.append(f)
            self._pending[i] = futs

    def batch(self, image_indices):
        import torch
        raw = self.raw_batch(image_indices)
        if not self.fused:
            return raw
        from . import runtime
        eng = runtime.engine()
        c0 = self.cm_list[0]
        out = torch.empty((raw.shape[0], len(self.cm_list), c0.height, c0.width, 3), dtype=torch.uint8,
                          device=self.device)
        for c, cm in enumerate(self.cm_list):
            eng.resample(cm, raw[:, c], out=out[:, c])
        return out

class RawDeviceFrameSource:
    """Raw (sensor-size) frames resident in HBM [F_total, C, H0, W0, 3]; every batch is undistort+resized on the
    device to the CameraManagers' output size (the reference does this per image on the host,
    cama/reproject.py:228-244)."""

    def __init__(self, raw, cm_list, fused=True):
        self.raw, self.cm_list = raw, cm_list
        self.fused = fused          # True: ClipManager.render_clip hands the raw frames to the fused overlay kernel
        self._buf = None

    def raw_batch(self, image_indices):
        """Raw frames of the given image indices: a view for a contiguous range (the usual case), a gathered copy for
        ranges with holes (clips whose pose lookup skips frames, cama/dataset.py:93-96)."""
        import torch
        idx = [int(i) for i in image_indices]
        if idx == list(range(idx[0], idx[0] + len(idx))):
            return self.raw[idx[0]:idx[0] + len(idx)]
        return self.raw.index_select(0, torch.as_tensor(idx, dtype=torch.long, device=self.raw.device))

    def batch(self, image_indices):
        import torch
        from . import runtime
        eng = runtime.engine()
        raw = self.raw_batch(image_indices)
        c0 = self.cm_list[0]
        shape = (int(raw.shape[0]), len(self.cm_list), c0.height, c0.width, 3)
        if self._buf is None or tuple(self._buf.shape) != shape:
            self._buf = torch.empty(shape, dtype=torch.uint8, device=self.raw.device)
        for c, cm in enumerate(self.cm_list):
            eng.resample(cm, raw[:, c], out=self._buf[:, c])
        return self._buf


def _pump_entry(ref, P):
    """Body of a ClipFrameSource's decode-pump thread.  `ref` is a weak reference to the source: while the thread is parked
    (waiting for the consumer to catch up) it holds no strong one, wakes up twice a second, and ends when the source is gone
    or the plan was cancelled."""
    try:
        for j, key in enumerate(P["keys"]):
            with P["cv"]:
                while True:
                    src = ref()
                    if src is None or P["stop"]:
                        return
                    depth = src._pump_depth
                    del src
                    if j < P["consumed"] + depth:
                        break
                    P["cv"].wait(timeout=0.5)
            src = ref()
            if src is None:
                return
            pend = src._pump_step(P, j, key)
            del src
            with P["cv"]:
                P["ready"][key] = pend
                P["cv"].notify_all()
            del pend
    except BaseException as e:                         # surfaces in the consumer's raw_batch()
        with P["cv"]:
            P["error"] = e
            P["cv"].notify_all()
