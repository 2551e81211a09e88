"""MI355X-native CAMA reprojection hot path.

Importing the package sets ONE process-wide default of the HIP runtime, and only when the caller has not chosen a value:
GPU_MAX_HW_QUEUES.  The runtime maps all streams of a process onto that many hardware queues (4 unless told otherwise), and two
streams that share one run their kernels strictly one after the other.  The device JPEG decoder splits a batch into seven
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
    torch = _sys.modules.get("torch")
    late = False
    try:
        late = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        late = False
    _os.environ["GPU_MAX_HW_QUEUES"] = str(HW_QUEUES)
    _hw_queues_note = ("GPU_MAX_HW_QUEUES=%d set after the HIP runtime was initialised: no effect in this process "
                       "(import cama / cama_amd before the first CUDA call)" % HW_QUEUES) if late else \
        "GPU_MAX_HW_QUEUES=%d set by cama_amd" % HW_QUEUES


def hw_queue_default():
    """What importing the package did about GPU_MAX_HW_QUEUES (a string for logs and bench lines)."""
    return _hw_queues_note


_set_hw_queue_default()
