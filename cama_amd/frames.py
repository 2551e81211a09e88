"""Camera-frame ingest for the overlay stage.

The reference reads one JPEG per camera per frame with cv2.imread and resamples it with
cv2.initUndistortRectifyMap + cv2.remap (cama/reproject.py:228-244).  OpenCV is installed on neither box
(SURVEY.md section 8c), so decode uses cv2 when importable, else Pillow (both wrap libjpeg; parity with
cv2.imread is unpinned), and `.npy` raw BGR frames are accepted next to / instead of the .jpg.

A FrameSource hands the overlay kernel a device tensor [F,C,H,W,3] uint8 BGR for a list of frame indices:
  * ClipFrameSource  -- reads <clip>/<camera>/<timestamp>.jpg|.npy (the demo's layout)
  * DeviceFrameSource -- frames already resident in HBM (bench / streaming pipelines)
"""
import os

import numpy as np


def read_bgr(path):
    """(H,W,3) uint8 BGR from a .jpg/.png path; falls back to the `.npy` twin of the same stem."""
    stem = os.path.splitext(path)[0]
    if not os.path.exists(path) and os.path.exists(stem + ".npy"):
        return np.ascontiguousarray(np.load(stem + ".npy"))
    if path.endswith(".npy"):
        return np.ascontiguousarray(np.load(path))
    try:
        import cv2
        img = cv2.imread(path)
        if img is None:
            raise FileNotFoundError(path)
        return img
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])


def resample_host_image(cm, image):
    """Undistort + resize one host image to the CameraManager's output size on the device."""
    import torch
    from . import runtime
    eng = runtime.engine()
    src = torch.from_numpy(np.ascontiguousarray(image)).to(eng.device)
    return eng.resample(cm, src).cpu().numpy()


class DeviceFrameSource:
    """Frames resident in HBM: tensor [F_total, C, H, W, 3] uint8; index i = sync image index i."""

    def __init__(self, frames, index_offset=0):
        self.frames = frames
        self.index_offset = index_offset

    def batch(self, image_indices):
        idx = [i - self.index_offset for i in image_indices]
        if len(idx) and idx == list(range(idx[0], idx[0] + len(idx))):
            return self.frames[idx[0]:idx[0] + len(idx)]          # contiguous view, no copy
        import torch
        return self.frames[torch.as_tensor(idx, device=self.frames.device)].contiguous()


class ClipFrameSource:
    """Decode + (if needed) resample + upload the demo's per-camera frame files."""

    def __init__(self, cm_list, device):
        self.cm_list = cm_list
        self.device = device

    def batch(self, image_indices):
        import torch
        c0 = self.cm_list[0]
        out = torch.empty((len(image_indices), len(self.cm_list), c0.height, c0.width, 3), dtype=torch.uint8,
                          device=self.device)
        for k, idx in enumerate(image_indices):
            for c, cm in enumerate(self.cm_list):
                img = cm.read_resized_image_by_index(idx)
                out[k, c].copy_(torch.from_numpy(np.ascontiguousarray(img)))
        return out
