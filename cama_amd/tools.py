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
import shutil
import subprocess

import numpy as np

MOSAIC_ORDER = ("camera_front_left", "camera_front", "camera_front_right",
                "camera_rear_left", "camera_rear", "camera_rear_right")   # tools.py:23-24
VIDEO_FPS = 10                                                            # tools.py:17


def load_json(filename):
    with open(filename, "r") as f:
        return json.load(f)


def _encoder_command(path, width, height, fps=VIDEO_FPS):
    return ["ffmpeg", "-loglevel", "quiet", "-y",
            "-f", "rawvideo", "-pix_fmt", "bgr24", "-s", f"{width}x{height}", "-i", "pipe:",
            "-pix_fmt", "yuv420p", "-vcodec", "libx264", "-r", str(fps), path]


class VideoGenerator:
    def __init__(self, output_video_path, output_shape=(2880, 1080)):
        self.output_video_path = output_video_path
        self.output_shape = tuple(output_shape)         # (width, height) of the mosaic
        if shutil.which("ffmpeg") is None:
            raise FileNotFoundError("the `ffmpeg` binary is needed to encode the reprojection video")
        self.writer = subprocess.Popen(_encoder_command(output_video_path, *self.output_shape),
                                       stdin=subprocess.PIPE)

    def concate_image(self, image_dict):
        """camera name -> (H,W,3) images => (2H,3W,3): front_left | front | front_right over the three rear cameras."""
        mosaic = getattr(image_dict, "mosaic", None)
        if callable(mosaic):
            ready = mosaic(MOSAIC_ORDER)               # already assembled in HBM by the overlay kernel
            if ready is not None:
                return ready
        rows = [np.concatenate([image_dict[name] for name in MOSAIC_ORDER[r:r + 3]], axis=1) for r in (0, 3)]
        return np.concatenate(rows, axis=0)

    def add_frame(self, image):
        """Raw BGR24 bytes of the mosaic into the encoder's pipe (tools.py:28-32 writes image.astype(uint8).tobytes():
        the same bytes; a contiguous uint8 array goes out through the buffer protocol without the two 9 MB copies)."""
        self.writer.stdin.write(memoryview(np.ascontiguousarray(image, dtype=np.uint8)).cast("B"))

    def add_frame_from_dict(self, image_dict):
        self.add_frame(self.concate_image(image_dict))

    def close(self):
        writer = getattr(self, "writer", None)
        if writer is not None:
            self.writer = None
            writer.stdin.close()
            writer.wait()

    def __del__(self):
        self.close()
