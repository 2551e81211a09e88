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
# egress: a VideoGenerator that can take planar YUV 4:2:0 announces itself here, and the render path then prepares
# every batch's I420 planes + download right behind its render (cama_amd/egress.py)
_egress_listeners = 0


def request_egress(mode):
    """mode "i420": one more listener (a VideoGenerator that takes planar YUV 4:2:0); None: one listener less.
    Counted, because main.py rebinds `vg = VideoGenerator(...)` per pass: the new object's __init__ runs BEFORE the old
    object's __del__ -> close(), and a single slot would leave the second video without the prefetch."""
    global _egress_listeners
    if mode == "i420":
        _egress_listeners += 1
    elif mode is None:
        _egress_listeners = max(0, _egress_listeners - 1)
    else:
        raise ValueError(f"unknown egress mode {mode!r}")


def egress_mode():
    return "i420" if _egress_listeners > 0 else None
