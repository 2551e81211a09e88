"""VideoGenerator / load_json with the reference's surface (cama/tools.py).

concate_image is the 2x3 mosaic the demo feeds to the encoder.  When render_vectors produced the frame on the
device, the mosaic already exists (the overlay kernel writes straight into mosaic addresses), so concate_image
only downloads it; for plain dicts of arrays it concatenates like the reference.

The encoder is outside the hot path.  The reference drives it through ffmpeg-python
(tools.py:13-20); here the same stream -- raw bgr24 frames on stdin -> libx264, yuv420p, 10 fps, overwrite,
quiet -- is a plain subprocess of the `ffmpeg` binary, started lazily so that importing this module and building
mosaics works on machines without ffmpeg.  `self.writer` keeps the `.stdin` / `.wait()` interface callers use.
"""
import json
import os
import shutil
import subprocess

import numpy as np

MOSAIC_ORDER = ("camera_front_left", "camera_front", "camera_front_right",
                "camera_rear_left", "camera_rear", "camera_rear_right")   # tools.py:23-24
VIDEO_FPS = 10                                                            # tools.py:17


def load_json(filename):
    with open(filename, "r") as f:
        return json.load(f)


def _encoder_command(path, width, height, fps=VIDEO_FPS, pix_fmt="bgr24"):
    return ["ffmpeg", "-loglevel", "quiet", "-y",
            "-f", "rawvideo", "-pix_fmt", pix_fmt, "-s", f"{width}x{height}", "-i", "pipe:",
            "-pix_fmt", "yuv420p", "-vcodec", "libx264", "-r", str(fps), path]


class _Sink:
    """Stands in for the encoder process when there is no ffmpeg (neither box has one) or a test wants the stream:
    `.stdin.write` goes to a file object, `.wait()` is a no-op."""

    def __init__(self, fileobj, close):
        self.stdin, self._close = fileobj, close

    def wait(self):
        return 0


class VideoGenerator:
    """cama/tools.py:12-40.  Same constructor, concate_image, add_frame.  The encoder (ffmpeg: rawvideo in -> libx264
    yuv420p out, 10 fps, overwrite, quiet) is started at the first add_frame, because only then it is known what
    crosses the pipe:

      * DEFAULT = the reference's bytes: concate_image returns an ndarray and add_frame pipes it as bgr24 rawvideo
        (tools.py:27-32).  For frames that come out of ClipManager.render_vectors the ndarray is a view of a pinned host
        copy of the whole render batch, downloaded asynchronously right behind the render (cama_amd/egress.py), so the
        per-frame loop never waits for a copy it could have started a batch earlier;
      * OPT-IN (CAMA_EGRESS=i420, or configs["egress"] = "i420" on the ClipManager): those frames leave the GPU as planar
        YUV 4:2:0 (cama_bgr_to_i420), the pipe carries `-pix_fmt yuv420p` rawvideo: half the bytes, no libswscale pass in
        ffmpeg.  The conversion restates libswscale's C arithmetic and could not be pinned against a real ffmpeg (none on
        any box; x86 builds may average chroma in a SIMD body), which is why it is not the default.

    `sink`: a binary file object that receives the raw stream instead of an encoder (tests; CAMA_VIDEO_SINK=null or a file
    path does the same for an unchanged main.py on a machine without ffmpeg)."""

    def __init__(self, output_video_path, output_shape=(2880, 1080), sink=None):
        self.output_video_path = output_video_path
        self.output_shape = tuple(output_shape)         # (width, height) of the mosaic
        self.writer = None
        self.pix_fmt = None                             # decided by the first frame
        self._sink = sink
        env = os.environ.get("CAMA_VIDEO_SINK")
        if sink is None and env:
            self._sink = open(os.devnull if env == "null" else env, "wb")
            self._own_sink = True
        else:
            self._own_sink = False
        if self._sink is None and shutil.which("ffmpeg") is None:
            raise FileNotFoundError("the `ffmpeg` binary is needed to encode the reprojection video")
        from . import runtime
        runtime.request_egress("listen")                # render batches prepare their host copies from now on
        self._asked_egress = True

    def _start(self, pix_fmt):
        self.pix_fmt = pix_fmt
        if self._sink is not None:
            self.writer = _Sink(self._sink, self._own_sink)
        else:
            self.writer = subprocess.Popen(_encoder_command(self.output_video_path, *self.output_shape, pix_fmt=pix_fmt),
                                           stdin=subprocess.PIPE)

    def concate_image(self, image_dict):
        """camera name -> (H,W,3) images => (2H,3W,3): front_left | front | front_right over the three rear cameras."""
        handle = getattr(image_dict, "mosaic_handle", None)
        if callable(handle):
            dev = handle(MOSAIC_ORDER)                 # already assembled in HBM by the overlay kernel
            if dev is not None:
                # default: the reference's return type to the letter, a plain ndarray (a view of the batch's pinned host
                # copy); with the i420 opt-in the mosaic stays in HBM behind an ndarray-like handle
                from . import runtime
                return dev if runtime.egress_format() == "i420" else dev.ndarray()
        mosaic = getattr(image_dict, "mosaic", None)
        if callable(mosaic):
            ready = mosaic(MOSAIC_ORDER)
            if ready is not None:
                return ready
        rows = [np.concatenate([image_dict[name] for name in MOSAIC_ORDER[r:r + 3]], axis=1) for r in (0, 3)]
        return np.concatenate(rows, axis=0)

    def add_frame(self, image):
        """One mosaic into the encoder's pipe (tools.py:28-32 writes image.astype(uint8).tobytes())."""
        pix_fmt = getattr(self, "pix_fmt", None) or ("bgr24" if getattr(self, "writer", None) is not None else None)
        i420 = getattr(image, "i420", None)
        planes = None
        if callable(i420) and pix_fmt in (None, "yuv420p"):
            from . import runtime
            if runtime.egress_format() == "i420":
                planes = i420()                         # pinned host bytes prepared behind the render; None = cannot
        if planes is not None:
            if self.writer is None:
                self._start("yuv420p")
            self.writer.stdin.write(memoryview(planes).cast("B"))
            return
        if getattr(self, "writer", None) is None:
            self._start("bgr24")
            pix_fmt = "bgr24"
        if pix_fmt == "yuv420p":                        # a plain array in an I420 stream: convert on the host
            from .egress import bgr_to_i420_host
            self.writer.stdin.write(memoryview(bgr_to_i420_host(np.asarray(image))).cast("B"))
            return
        # the same bytes as the reference; a contiguous uint8 array goes out through the buffer protocol
        self.writer.stdin.write(memoryview(np.ascontiguousarray(image, dtype=np.uint8)).cast("B"))

    def add_frame_from_dict(self, image_dict):
        self.add_frame(self.concate_image(image_dict))

    def close(self):
        if getattr(self, "_asked_egress", False):       # nobody is listening any more
            self._asked_egress = False
            from . import runtime
            runtime.request_egress(None)
        writer = getattr(self, "writer", None)
        if writer is not None:
            self.writer = None
            if not isinstance(writer, _Sink):
                writer.stdin.close()
                writer.wait()
            elif writer._close:
                writer.stdin.close()
        elif getattr(self, "_own_sink", False) and self._sink is not None:
            self._sink.close()
        self._sink = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
