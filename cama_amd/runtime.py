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
_egress = None


def request_egress(mode):
    """mode: "i420" or None."""
    global _egress
    _egress = mode


def egress_mode():
    return _egress
