"""VideoGenerator / load_json with the reference's surface (cama/tools.py).

concate_image is the 2x3 mosaic the demo feeds to the encoder.  When render_vectors produced the frame on the
device, the mosaic already exists (the overlay kernel writes straight into mosaic addresses), so
concate_image only downloads it; for plain dicts of arrays it concatenates like the reference.
The ffmpeg pipe itself is outside the hot path; it is created lazily so that importing this module (and
building mosaics) works where ffmpeg-python is not installed.
"""
import json

import numpy as np

MOSAIC_ORDER = ("camera_front_left", "camera_front", "camera_front_right",
                "camera_rear_left", "camera_rear", "camera_rear_right")   # tools.py:23-24


def load_json(filename):
    with open(filename, "r") as f:
        return json.load(f)


class VideoGenerator:
    def __init__(self, output_video_path, output_shape=(2880, 1080)):
        import ffmpeg   # ffmpeg-python, as in the reference (tools.py:13-20)
        self.writer = (
            ffmpeg
            .input('pipe:', format='rawvideo', pix_fmt='bgr24', s=f'{output_shape[0]}x{output_shape[1]}')
            .output(output_video_path, pix_fmt='yuv420p', vcodec='libx264', r=10, loglevel='quiet')
            .overwrite_output()
            .run_async(pipe_stdin=True)
        )

    def concate_image(self, image_dict):
        mosaic = getattr(image_dict, "mosaic", None)
        if callable(mosaic):
            got = mosaic(MOSAIC_ORDER)
            if got is not None:
                return got
        top = np.concatenate([image_dict[n] for n in MOSAIC_ORDER[:3]], axis=1)
        bottom = np.concatenate([image_dict[n] for n in MOSAIC_ORDER[3:]], axis=1)
        return np.concatenate([top, bottom], axis=0)

    def add_frame(self, image):
        self.writer.stdin.write(image.astype(np.uint8).tobytes())

    def add_frame_from_dict(self, image_dict):
        self.add_frame(self.concate_image(image_dict))

    def __del__(self):
        writer = getattr(self, "writer", None)
        if writer is not None:
            writer.stdin.close()
            writer.wait()
