"""Host-side behaviour of the egress helpers (no GPU): the listener count behind runtime.egress_mode() and the
ndarray-likeness of egress.DeviceMosaic (what VideoGenerator.concate_image hands to main.py's loop, cama/tools.py:22-32)."""
import io

import numpy as np

from cama_amd import egress, runtime
from cama_amd.tools import VideoGenerator


class _FakeBatch:
    """Stands in for egress.RenderBatch: `mosaic` only needs a shape; bgr() / i420() count their calls."""

    def __init__(self, arr):
        self._arr = arr
        self.mosaic = arr                        # [B, 2H, 3W, 3]
        self.bgr_calls = self.i420_calls = 0

    def bgr(self, j):
        self.bgr_calls += 1
        return self._arr[j].copy()

    def i420(self, j):
        self.i420_calls += 1
        return egress.bgr_to_i420_host(self._arr[j])


def test_egress_listeners_are_counted_across_a_rebind():
    """main.py rebinds `vg = VideoGenerator(...)` per dataset pass: the new generator's __init__ runs before the old one's
    __del__ -> close().  The second generator must still find the host-copy prefetch switched on.  The format is the
    reference's bgr24 unless CAMA_EGRESS / configs["egress"] opt into I420."""
    assert runtime.egress_mode() is None
    vg = VideoGenerator("a.mp4", sink=io.BytesIO())
    assert runtime.egress_mode() == "bgr24"
    old = vg
    vg = VideoGenerator("b.mp4", sink=io.BytesIO())      # (the rebind: `old` is closed only afterwards)
    old.close()
    assert runtime.egress_mode() == "bgr24"
    old.close()                                            # closing twice releases once
    assert runtime.egress_mode() == "bgr24"
    runtime.set_egress_format("i420")                      # what ClipManager does for configs["egress"] = "i420"
    try:
        assert runtime.egress_mode() == "i420"
    finally:
        runtime.set_egress_format(None)
    vg.close()
    assert runtime.egress_mode() is None


def test_egress_format_comes_from_the_environment_or_the_configs(monkeypatch):
    import pytest
    assert runtime.egress_format() == "bgr24"              # default = the reference's stream
    monkeypatch.setenv("CAMA_EGRESS", "i420")
    assert runtime.egress_format() == "i420"
    runtime.set_egress_format("bgr24")                     # an explicit choice beats the environment
    try:
        assert runtime.egress_format() == "bgr24"
    finally:
        runtime.set_egress_format(None)
    monkeypatch.setenv("CAMA_EGRESS", "rgb565")
    with pytest.raises(ValueError):
        runtime.egress_format()
    with pytest.raises(ValueError):
        runtime.set_egress_format("rgb565")


def test_device_mosaic_behaves_like_the_ndarray_once_touched():
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (2, 4, 32, 3), dtype=np.uint8)
    b = _FakeBatch(arr)
    m = egress.DeviceMosaic(b, 1)
    assert m.shape == (4, 32, 3) and m.dtype == np.uint8 and len(m) == 4 and b.bgr_calls == 0
    assert m.i420() is not None and b.bgr_calls == 0       # untouched: the device-side planes
    c = m.copy()                                           # ndarray methods come from the downloaded array
    assert isinstance(c, np.ndarray) and np.array_equal(c, arr[1]) and b.bgr_calls == 1
    assert m.reshape(-1).shape == (4 * 32 * 3,) and m.mean() == arr[1].mean() and m.size == arr[1].size
    m[0, 0] = (1, 2, 3)                                    # in-place edit, like drawing on the frame
    assert tuple(np.asarray(m)[0, 0]) == (1, 2, 3) and b.bgr_calls == 1
    assert m.i420() is None                                # the edited host bytes are the frame now
    assert m.tobytes() == np.asarray(m).tobytes() and m.astype(np.uint8).tobytes() == m.tobytes()


def test_add_frame_writes_the_edited_host_bytes_into_an_i420_stream(monkeypatch):
    monkeypatch.setenv("CAMA_EGRESS", "i420")              # the opt-in
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, (2, 4, 32, 3), dtype=np.uint8)
    b = _FakeBatch(arr)
    sink = io.BytesIO()
    vg = VideoGenerator("x.mp4", output_shape=(32, 4), sink=sink)
    try:
        vg.add_frame(egress.DeviceMosaic(b, 0))            # untouched -> the prepared planes, stream becomes yuv420p
        assert vg.pix_fmt == "yuv420p"
        touched = egress.DeviceMosaic(b, 1)
        touched[:, :8] = 0
        vg.add_frame(touched)                              # touched -> converted on the host from the EDITED pixels
        want = np.concatenate([egress.bgr_to_i420_host(arr[0]), egress.bgr_to_i420_host(np.asarray(touched))])
        assert sink.getvalue() == want.tobytes()
    finally:
        vg.close()


def test_concate_image_returns_a_plain_ndarray_by_default_and_the_stream_is_bgr24(monkeypatch):
    rng = np.random.default_rng(2)
    arr = rng.integers(0, 256, (1, 4, 32, 3), dtype=np.uint8)
    batch = _FakeBatch(arr)

    class Frame(dict):
        def mosaic_handle(self, order=None):
            return egress.DeviceMosaic(batch, 0)
    monkeypatch.delenv("CAMA_EGRESS", raising=False)
    sink = io.BytesIO()
    vg = VideoGenerator("x.mp4", output_shape=(32, 4), sink=sink)
    try:
        assert runtime.egress_mode() == "bgr24"
        out = vg.concate_image(Frame())
        assert type(out) is np.ndarray and out.shape == (4, 32, 3) and np.array_equal(out, arr[0])
        vg.add_frame(out)
        assert vg.pix_fmt == "bgr24" and sink.getvalue() == arr[0].tobytes() and batch.i420_calls == 0
    finally:
        vg.close()
    # opt-in: the handle stays a DeviceMosaic and the stream is yuv420p
    monkeypatch.setenv("CAMA_EGRESS", "i420")
    sink = io.BytesIO()
    vg = VideoGenerator("y.mp4", output_shape=(32, 4), sink=sink)
    try:
        out = vg.concate_image(Frame())
        assert isinstance(out, egress.DeviceMosaic)
        vg.add_frame(out)
        assert vg.pix_fmt == "yuv420p" and sink.getvalue() == egress.bgr_to_i420_host(arr[0]).tobytes()
    finally:
        vg.close()


def test_pinned_pool_keeps_a_buffer_out_of_circulation_while_views_of_it_live():
    """bgr24 mode hands out ndarray VIEWS of a batch's pinned host copy; the pool must not give that buffer to the next
    batch while a caller still holds one (main.py's `image` outlives the batch by one loop iteration).  The buffers are
    torch tensors and the views NON-OWNING arrays (`tensor.numpy()`): numpy makes the root `.numpy()` array the base of
    every derived view, so that root -- not a reshape of it -- is what the pool has to watch."""
    import torch
    pool = egress.PinnedPool()
    pool.alloc = lambda n: torch.empty(n, dtype=torch.uint8)       # (no GPU here: pageable stands in for pinned)
    buf = torch.zeros(64, dtype=torch.uint8)
    root = buf.numpy()
    rows = root.reshape(4, 16)
    view = rows[2]
    assert view.base is root and rows.base is root
    pool.give(buf, root)
    del rows, root
    pool._sweep()
    assert pool.free == {} and len(pool.limbo) == 1        # a view is out: parked
    assert pool.take(64) is not buf                        # ... and NOT handed to the next batch
    view[:] = 7                                            # (still the caller's bytes)
    del view
    pool._sweep()
    assert pool.free[64] == [buf] and not pool.limbo       # view gone: back in circulation
    # a derived (reshaped) array passed by mistake is resolved to its root
    buf2 = torch.zeros(32, dtype=torch.uint8)
    rows2 = buf2.numpy().reshape(2, 16)
    held = rows2[1]
    pool.give(buf2, rows2)
    del rows2
    pool._sweep()
    assert 32 not in pool.free and len(pool.limbo) == 1
    del held
    pool._sweep()
    assert len(pool.free[32]) == 1
    pool.give(torch.zeros(16, dtype=torch.uint8))          # nothing went out at all (I420 planes are copied into the pipe)
    assert len(pool.free[16]) == 1


def test_render_batch_frames_survive_the_batch_and_the_next_take(monkeypatch):
    """The sequence the advisor reproduced: a frame handed out by RenderBatch.bgr() is held, the batch dies, the pool is
    swept and asked for a buffer of the same size -- the held frame must keep its bytes."""
    import torch

    class FakeEvent:
        def synchronize(self):
            pass
    pool = egress.PinnedPool()
    pool.alloc = lambda n: torch.empty(n, dtype=torch.uint8)
    monkeypatch.setattr(egress, "_POOL", pool)
    b = egress.RenderBatch.__new__(egress.RenderBatch)
    b.engine, b.ids = None, [1, 2]
    b.mosaic = torch.zeros((2, 2, 4, 3), dtype=torch.uint8)
    b._host = torch.arange(48, dtype=torch.uint8)
    b._event, b._i420_dev, b._host_np, b._host_root, b._fmt = FakeEvent(), None, None, None, "bgr24"
    frame = b.bgr(1)
    want = frame.copy()
    host = b._host
    del b
    import gc
    gc.collect()
    nxt = pool.take(48)
    assert nxt is not host                                  # the held frame's buffer is not recycled
    nxt.fill_(255)
    assert np.array_equal(frame, want)
    del frame
    assert pool.take(48) is host
