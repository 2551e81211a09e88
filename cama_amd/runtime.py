"""Process-wide Engine (one process per GPU).  The device is cuda:$LOCAL_RANK unless CAMA_DEVICE says otherwise."""
import os

_engine = None


def default_device():
    dev = os.environ.get("CAMA_DEVICE")
    if dev:
        return dev
    return f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"


def engine():
    """The shared Engine; raises cama_amd._lib.CamaHipError without a GPU or without libcama_hip.so."""
    global _engine
    if _engine is None:
        from .engine import Engine
        _engine = Engine(default_device())
    return _engine


def set_engine(e):
    global _engine
    _engine = e


# ---------------------------------------------------------------------------------------------------------------
# egress: a VideoGenerator announces itself here, and the render path then prepares every batch's host copy right behind
# its render (cama_amd/egress.py) -- an asynchronous download into pinned memory, so that the per-frame loop of main.py
# never waits for a device->host copy it could have started a batch earlier.
#   "bgr24" (default)  the reference's bytes: concate_image returns an ndarray, add_frame pipes bgr24 (cama/tools.py:13-32)
#   "i420"  (opt-in: CAMA_EGRESS=i420 or configs["egress"] = "i420")  the mosaic leaves the GPU as planar YUV 4:2:0, the
#           encoder's own pixel format: half the bytes; the conversion restates libswscale's C path and is unpinned (no
#           ffmpeg on any box), which is why it is not the default
_egress_listeners = 0
_egress_format = None               # None = from the environment


def egress_format():
    """The stream format a VideoGenerator asks for: "bgr24" unless CAMA_EGRESS / set_egress_format say "i420"."""
    fmt = _egress_format or os.environ.get("CAMA_EGRESS", "bgr24")
    if fmt not in ("bgr24", "i420"):
        raise ValueError(f"unknown egress format {fmt!r} (bgr24 | i420)")
    return fmt


def set_egress_format(fmt):
    """configs["egress"] of a ClipManager lands here; None = back to the environment's choice."""
    global _egress_format
    if fmt not in (None, "bgr24", "i420"):
        raise ValueError(f"unknown egress format {fmt!r} (bgr24 | i420)")
    _egress_format = fmt


def request_egress(mode):
    """mode "listen" (or, from older callers, a format name): one more listener (a VideoGenerator); None: one listener
    less.  Counted, because main.py rebinds `vg = VideoGenerator(...)` per pass: the new object's __init__ runs BEFORE the
    old object's __del__ -> close(), and a single slot would leave the second video without the prefetch."""
    global _egress_listeners
    if mode in ("listen", "i420", "bgr24"):
        _egress_listeners += 1
    elif mode is None:
        _egress_listeners = max(0, _egress_listeners - 1)
    else:
        raise ValueError(f"unknown egress mode {mode!r}")


def egress_mode():
    """The format render batches should prepare on the host side, or None when nobody is listening."""
    return egress_format() if _egress_listeners > 0 else None
