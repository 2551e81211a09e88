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
    batch while a caller still holds one (main.py's `image` outlives the batch by one loop iteration)."""
    class Buf:                                             # stands in for a pinned torch tensor
        def __init__(self, n):
            self.n = n

        def numel(self):
            return self.n
    pool = egress.PinnedPool()
    buf, rows = Buf(64), np.zeros((4, 16), np.uint8)
    view = rows[2]
    pool.give(buf, rows)
    del rows
    pool._sweep()
    assert pool.free == {} and len(pool.limbo) == 1        # a view is out: parked
    del view
    pool._sweep()
    assert pool.free[64] == [buf] and not pool.limbo       # view gone: back in circulation
    rows2 = np.zeros((4, 16), np.uint8)
    pool.give(Buf(32), rows2)                              # no view handed out: back as soon as the batch lets go of it
    del rows2
    pool._sweep()
    assert len(pool.free[32]) == 1
    pool.give(Buf(16))                                     # nothing went out at all (I420 planes are copied into the pipe)
    assert len(pool.free[16]) == 1
