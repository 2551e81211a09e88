"""Render-ahead batches and mosaic egress for the reference's frame-by-frame loop (main.py:57-61).

The loop asks for one frame at a time; on the device one frame is ~10 us of work behind ~150 us of host issue cost, and
what leaves the GPU per frame is a 9.3 MB BGR mosaic that the encoder's front end immediately converts to yuv420p
(cama/tools.py:13-20).  So the drop-in classes work in batches behind the unchanged per-frame surface:

  * ClipManager.render_vectors renders `render_ahead` consecutive frames of the pass in ONE launch the first time a frame
    of the batch is asked for (all poses of a pass are known up front) and hands out slices afterwards; the next batch
    is issued one batch early, so the GPU works while the host consumes.
  * when a VideoGenerator is listening every batch is copied to pinned host memory asynchronously, right behind its
    render: as the BGR mosaics themselves (default: the reference's bgr24 stream, concate_image returns an ndarray view
    of that copy), or -- opt-in, runtime.egress_format() == "i420" -- converted to planar YUV 4:2:0 on the device first
    (cama_bgr_to_i420: half the bytes).  The per-frame add_frame() is then a wait on an already finished copy plus a
    pipe write.
"""
import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


class PinnedPool:
    """Pinned host buffers are expensive to create (hipHostMalloc): recycle them by size.  In bgr24 mode concate_image
    hands out ndarray VIEWS of a batch's buffer: such a buffer comes back with the array those views are slices of and
    waits in `limbo` until that array is gone (a view keeps its base alive, so a dead weak reference = no view left)."""

    def __init__(self):
        self.free = {}
        self.limbo = []                 # (buffer, weak reference to the ndarray every handed-out view is a view of)

    def _sweep(self):
        keep = []
        for buf, ref in self.limbo:
            if ref() is None:
                self._shelve(buf)
            else:
                keep.append((buf, ref))
        self.limbo = keep

    def _shelve(self, buf):
        lst = self.free.setdefault(buf.numel(), [])
        if len(lst) < 4:
            lst.append(buf)

    def take(self, nbytes):
        if self.limbo:
            self._sweep()
        lst = self.free.get(nbytes)
        if lst:
            return lst.pop()
        return self.alloc(nbytes)

    @staticmethod
    def alloc(nbytes):
        return _torch().empty(nbytes, dtype=_torch().uint8, pin_memory=True)

    def give(self, buf, views_of=None):
        """Hand a buffer back.  views_of: the ROOT ndarray of the views that went out to callers -- the array whose `.base`
        is not an ndarray, i.e. what `tensor.numpy()` returned: numpy makes that the base of every derived view, however
        many reshapes / slices lie in between -- or None: nothing went out."""
        if views_of is not None and isinstance(getattr(views_of, "base", None), np.ndarray):
            views_of = views_of.base                # a derived view was passed: watch its root instead
        if views_of is None:
            self._shelve(buf)
            return
        import weakref
        if len(self.limbo) >= 8:                # (callers hoarding frames: let the oldest go, the GC frees it)
            self.limbo.pop(0)
        self.limbo.append((buf, weakref.ref(views_of)))


_POOL = PinnedPool()
class RenderBatch:
    """B consecutive frames rendered in one launch: `mosaic` [B, rows*H, cols*W, 3] uint8 in HBM (complete on the
    stream it was rendered on), plus -- after start_egress() -- their host copy on its way to pinned memory: the BGR
    bytes themselves ("bgr24", the reference's stream) or their I420 planes ("i420")."""

    def __init__(self, engine, image_ids, mosaic):
        self.engine, self.ids, self.mosaic = engine, [int(i) for i in image_ids], mosaic
        self._host = self._event = self._i420_dev = None
        self._host_np = self._host_root = None
        self._fmt = None

    def start_egress(self, fmt="bgr24"):
        """Enqueue the download (i420: BGR -> I420 first) behind the render, on the same stream.  False when the mosaic
        shape does not fit the I420 converter (odd height, width not a multiple of 16): the caller then falls back to BGR."""
        torch = _torch()
        B, H2, W2 = (int(v) for v in self.mosaic.shape[:3])
        if self._event is not None:
            return self._fmt == fmt
        eng = self.engine
        if fmt == "bgr24":
            if not self.mosaic.is_contiguous():
                return False
            per = H2 * W2 * 3
            with torch.cuda.device(eng.device):
                self._host = _POOL.take(B * per)
                self._download(self._host.view(B, per), self.mosaic.view(B, per), self.mosaic)
            self._fmt = fmt
            return True
        per = H2 * W2 * 3 // 2
        if H2 % 2 or W2 % 16 or per % 16 or not self.mosaic.is_contiguous():
            return False
        with torch.cuda.device(eng.device):
            self._i420_dev = torch.empty((B, per), dtype=torch.uint8, device=eng.device)
            _lib.check(eng.lib.cama_bgr_to_i420(self.mosaic.data_ptr(), H2 * W2 * 3, self._i420_dev.data_ptr(), per, B, H2,
                                                W2, eng._stream()))
            self._host = _POOL.take(B * per)
            self._download(self._host.view(B, per), self._i420_dev, self._i420_dev)
        self._fmt = "i420"
        return True

    def _download(self, host_view, dev_view, keep):
        """Asynchronous device -> pinned copy behind the work queued on the current stream; records self._event.  (A CU-masked
        side stream for the copies was tried in round 4 and measured slower: profiles/r04_demo_loop_timeline.txt.)"""
        torch = _torch()
        cur = torch.cuda.current_stream(self.engine.device)
        self._event = torch.cuda.Event()
        host_view.copy_(dev_view, non_blocking=True)
        self._event.record(cur)

    def _host_rows(self):
        if self._host_np is None:
            self._event.synchronize()
            B = int(self.mosaic.shape[0])
            # numpy collapses the base chain of a view to the ROOT ndarray: every frame view handed out by bgr() has the
            # array `.numpy()` returns as its base, not the reshaped one -- that root is what the pool has to watch
            self._host_root = self._host.numpy()
            self._host_np = self._host_root.reshape(B, -1)
            self._i420_dev = None
        return self._host_np

    def i420(self, j):
        """The I420 bytes of frame j as a uint8 ndarray view of the pinned buffer (valid while the batch lives)."""
        if self._event is None and not self.start_egress("i420"):
            return None
        if self._fmt != "i420":
            return None
        return self._host_rows()[j]

    def bgr(self, j):
        """Frame j's mosaic (2H, 3W, 3) as a host ndarray: a view of the batch's pinned copy when the bgr24 download was
        started behind the render (the buffer is not reused while the view lives), else a synchronous download."""
        if self._fmt == "bgr24":
            return self._host_rows()[j].reshape(tuple(int(v) for v in self.mosaic.shape[1:]))
        return self.mosaic[j].cpu().numpy()

    def __del__(self):
        host = getattr(self, "_host", None)
        if host is not None:
            ev = getattr(self, "_event", None)
            try:
                if ev is not None:
                    ev.synchronize()            # the copy into it must be over before someone else reuses it
                root = self._host_root
                self._host_np = self._host_root = None
                _POOL.give(host, root)
            except Exception:
                pass


class DeviceMosaic:
    """What VideoGenerator.concate_image returns on the fused path: the (2H, 3W, 3) uint8 BGR mosaic, still in HBM.
    Behaves like the ndarray the reference returns when touched (np.asarray / astype / tobytes / indexing download it
    once); VideoGenerator.add_frame takes the I420 shortcut instead."""

    def __init__(self, batch, j):
        self.batch, self.j = batch, j
        self.shape = tuple(int(v) for v in batch.mosaic.shape[1:])
        self.dtype = np.dtype(np.uint8)
        self.ndim = 3
        self._bgr = None

    def ndarray(self):
        """The reference's return value of concate_image: the mosaic as a real ndarray (a view of the batch's pinned
        host copy when the bgr24 download was started behind the render)."""
        return self._host()

    def _host(self):
        """The mosaic as a host ndarray (downloaded once).  From here on this object IS that array for every purpose:
        the caller may have edited it in place, so the device-side I420 shortcut is off (i420() -> None) and
        VideoGenerator.add_frame converts / writes the host bytes."""
        if self._bgr is None:
            self._bgr = self.batch.bgr(self.j)
        return self._bgr

    def __array__(self, dtype=None, copy=None):
        a = self._host()
        return a if dtype is None else a.astype(dtype, copy=False)

    def astype(self, dtype, **kw):
        return self._host().astype(dtype, **kw)

    def tobytes(self):
        return self._host().tobytes()

    def __getitem__(self, k):
        return self._host()[k]

    def __setitem__(self, k, v):
        self._host()[k] = v

    def __len__(self):
        return self.shape[0]

    def __getattr__(self, name):
        # everything else an ndarray offers (copy, reshape, mean, size, T, ...) comes from the downloaded array
        if name.startswith("__") or name in ("batch", "j", "_bgr"):
            raise AttributeError(name)
        return getattr(self._host(), name)

    def i420(self):
        if self._bgr is not None:               # touched on the host: those bytes are the frame now
            return None
        return self.batch.i420(self.j)


def bgr_to_i420_host(image):
    """Host twin of cama_bgr_to_i420 for frames that reach an I420-mode VideoGenerator as plain ndarrays: libswscale's
    unscaled BGR24 -> YUV420P C arithmetic (BT.601 limited range, 15-bit fixed point, chroma from the first pixel of
    each 2x2 block).  Returns the planar bytes as one uint8 vector."""
    img = np.asarray(image, np.uint8)
    H, W = img.shape[:2]
    if H % 2 or W % 2:
        raise ValueError("yuv420p needs even frame sizes")
    b, g, r = (img[..., k].astype(np.int32) for k in range(3))
    y = ((8414 * r + 16519 * g + 3208 * b) >> 15) + 16
    b0, g0, r0 = b[0::2, 0::2], g[0::2, 0::2], r[0::2, 0::2]
    u = ((-4865 * r0 - 9528 * g0 + 14392 * b0) >> 15) + 128
    v = ((14392 * r0 - 12061 * g0 - 2332 * b0) >> 15) + 128
    return np.concatenate([y.astype(np.uint8).reshape(-1), u.astype(np.uint8).reshape(-1), v.astype(np.uint8).reshape(-1)])
