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
    __del__ -> close().  The second generator must still find the I420 prefetch switched on."""
    assert runtime.egress_mode() is None
    vg = VideoGenerator("a.mp4", sink=io.BytesIO())
    assert runtime.egress_mode() == "i420"
    old = vg
    vg = VideoGenerator("b.mp4", sink=io.BytesIO())      # (the rebind: `old` is closed only afterwards)
    old.close()
    assert runtime.egress_mode() == "i420"
    old.close()                                            # closing twice releases once
    assert runtime.egress_mode() == "i420"
    vg.close()
    assert runtime.egress_mode() is None


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


def test_add_frame_writes_the_edited_host_bytes_into_an_i420_stream():
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


def test_concate_image_returns_a_plain_ndarray_in_bgr24_mode(monkeypatch):
    arr = np.zeros((1, 4, 32, 3), np.uint8)

    class Frame(dict):
        def mosaic_handle(self, order=None):
            return egress.DeviceMosaic(_FakeBatch(arr), 0)
    monkeypatch.setenv("CAMA_EGRESS", "bgr24")
    vg = VideoGenerator("x.mp4", sink=io.BytesIO())
    try:
        assert runtime.egress_mode() is None               # bgr24 mode never asks for the I420 prefetch
        out = vg.concate_image(Frame())
        assert type(out) is np.ndarray and out.shape == (4, 32, 3)
    finally:
        vg.close()
