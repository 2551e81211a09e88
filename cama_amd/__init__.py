"""MI355X-native CAMA reprojection hot path.

Importing the package sets ONE process-wide default of the HIP runtime -- GPU_MAX_HW_QUEUES -- and only when (a) the caller
has not chosen a value, (b) the HIP runtime is not initialised yet (afterwards the variable has no effect: it is left alone and
hw_queue_default() says so) and (c) the device JPEG decoder can be used at all (CAMA_JPEG_DECODER=host: nothing is set).
The runtime maps all streams of a process onto that many hardware queues (4 unless told otherwise), and two streams that share one run their kernels strictly one after the other.  The device JPEG decoder splits a batch into seven
groups on seven streams so that their latency-bound entropy chains tile the chip; on four queues only four of them were ever
in flight (profiles/r05_jpeg_decoder.txt: 240 photo-like 1600x900 frames 51.2 k -> 57.3 k images/s with eight, same box, same
library).  The variable is read when the runtime initialises, i.e. at the first HIP call of the process: `import cama` /
`import cama_amd` at the top of a script (main.py does that) is early enough; `hw_queue_default()` says what happened.
"""
import os as _os
import sys as _sys

HW_QUEUES = 8
_hw_queues_note = None


def _set_hw_queue_default():
    global _hw_queues_note
    if "GPU_MAX_HW_QUEUES" in _os.environ:
        _hw_queues_note = "caller's GPU_MAX_HW_QUEUES=%s kept" % _os.environ["GPU_MAX_HW_QUEUES"]
        return
    if _os.environ.get("CAMA_JPEG_DECODER") == "host":
        _hw_queues_note = "GPU_MAX_HW_QUEUES left alone (CAMA_JPEG_DECODER=host: only the device decoder's groups need the queues)"
        return
    torch = _sys.modules.get("torch")
    late = False
    try:
        late = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        late = False
    if late:
        _hw_queues_note = ("GPU_MAX_HW_QUEUES left alone: the HIP runtime was initialised before cama_amd was imported, the "
                           "variable would have no effect (import cama / cama_amd before the first CUDA call to let the device "
                           "JPEG decoder's %d groups run side by side)" % HW_QUEUES)
        return
    _os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES)
    _hw_queues_note = "GPU_MAX_HW_QUEUES=%d set by cama_amd" % HW_QUEUES


def hw_queue_default():
    """What importing the package did about GPU_MAX_HW_QUEUES (a string for logs and bench lines)."""
    return _hw_queues_note


_set_hw_queue_default()
