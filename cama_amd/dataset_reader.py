"""Clip ("pack") reader with the reference's DatasetReader surface (cama/dataset_reader.py): attribute.json, sensor
timestamps, the calibration graph, intrinsics, odometry text files, sensor file paths -- what the reprojection path
touches -- plus the per-sensor iterators (lidar / IMU / GNSS / wheel / camera / semantic) and the GNSS / wheel ->
TUM converters.  Pure host-side file parsing, once per clip.  Images are decoded with Pillow (no OpenCV in this
image) and returned in OpenCV's channel order (BGR / BGRA).
"""
import json
import os
from collections import deque

import numpy as np

from .pose_transformer import invT


class DatasetReader:
    def __init__(self, pack_path=None):
        self.attribute = dict()
        self.extrinsic_graph = None
        self.pack_path = ""
        if pack_path:
            self.read_pack(pack_path)

    def read_pack(self, path):
        """Load <path>/attribute.json; FileNotFoundError if absent (dataset_reader.py:19-37)."""
        self.pack_path = path
        attribute_path = os.path.join(self.pack_path, "attribute.json")
        if not os.path.exists(attribute_path):
            raise FileNotFoundError("can not find {}".format(attribute_path))
        with open(attribute_path, "r") as f:
            self.attribute = json.load(f)

    # ------------------------------------------------------------------ timestamps / files
    def get_sensor_timestamp(self, sensor_name, sync=True):
        """ms integers -> list of python floats in seconds (dataset_reader.py:39-43)."""
        stamps = np.asarray(self.attribute["sync" if sync else "unsync"][sensor_name]).astype(np.double)
        stamps /= 1000.0
        return stamps.tolist()

    def yield_sensor_filepath(self, sensor_name, ext, sync=True, start_idx=None, end_idx=None,
                              start_time=None, end_time=None):
        """Absolute paths <pack>/<sensor>/<timestamp_ms>.<ext>, by index range or by time range
        (dataset_reader.py:100-148)."""
        stamps_ms = self.attribute["sync" if sync else "unsync"][sensor_name]
        secs = np.asarray(stamps_ms) / 1000.
        if start_time is not None or end_time is not None:
            if start_time is None or start_time <= secs[0]:
                start_idx = None
            elif start_time > secs[-1]:
                start_idx = -1
            else:
                start_idx = np.searchsorted(secs, start_time, side="left")
            if end_time is None or end_time >= secs[-1]:
                end_idx = None
            elif end_time < secs[0]:
                end_idx = -1
            else:
                end_idx = np.searchsorted(secs, end_time, side="left") - 1
            if (start_idx is not None and start_idx < 0) or (end_idx is not None and end_idx < 0):
                stamps_ms = []
        for ts in stamps_ms[start_idx:end_idx] if len(stamps_ms) else []:
            yield os.path.join(self.pack_path, sensor_name, "{}.{}".format(ts, ext))

    # ------------------------------------------------------------------ calibration graph
    def _edge(self, src, dst):
        """Direct or inverted calibration entry, else None (dataset_reader.py:150-168)."""
        if src == dst:
            return np.eye(4, dtype=np.float32)
        cal = self.attribute["calibration"]
        fwd = "{}_2_{}".format(src, dst)
        if fwd in cal:
            return np.asarray(cal[fwd])
        bwd = "{}_2_{}".format(dst, src)
        if bwd in cal:
            return invT(np.asarray(cal[bwd]))
        return None

    def _build_graph(self):
        graph = {}
        for key in self.attribute["calibration"]:
            if "_2_" in key:
                a, b = key.split("_2_")
                graph.setdefault(a, []).append(b)
                graph.setdefault(b, []).append(a)
        self.extrinsic_graph = graph

    def get_extrinsic_path(self, from_sensor, to_sensor):
        """Breadth-first shortest chain of sensors linking the two (dataset_reader.py:181-220)."""
        if self.extrinsic_graph is None:
            self._build_graph()
        if from_sensor == to_sensor:
            return None
        done = set()
        frontier = deque([[from_sensor]])
        while frontier:
            chain = frontier.popleft()
            node = chain[-1]
            if node in done:
                continue
            for nxt in self.extrinsic_graph.get(node, []):
                longer = chain + [nxt]
                if nxt == to_sensor:
                    return longer
                frontier.append(longer)
            done.add(node)
        return None

    def get_extrinsic(self, from_sensor, to_sensor):
        """4x4 that maps a point in `from_sensor` coordinates to `to_sensor` coordinates
        (dataset_reader.py:222-248): direct entry, inverse entry, else product along the BFS chain."""
        T = self._edge(from_sensor, to_sensor)
        if T is not None:
            return T
        chain = self.get_extrinsic_path(from_sensor, to_sensor)
        if chain is None:
            print("extrinsic path not found!")
            return None
        T = np.eye(4, dtype=np.float32)
        for a, b in zip(chain[:-1], chain[1:]):
            T = self._edge(a, b) @ T
        return T

    def get_all_sensors(self):
        names = set()
        for key in self.attribute["calibration"]:
            names.update(key.split("_2_"))
        return list(names)

    # ------------------------------------------------------------------ intrinsics / odometry
    def get_intrinsic(self, sensor):
        from warnings import warn
        warn("get_intrinsic() is deprecated, use get_intrinsics() instead")
        entry = self.attribute["calibration"][sensor]
        return np.asarray(entry["K"]), np.asarray(entry["d"])

    def get_intrinsics(self, sensor):
        """dict(K, d, width, height, hfov) (dataset_reader.py:278-294)."""
        entry = self.attribute["calibration"][sensor]
        return {"K": np.asarray(entry.get("K", None)), "d": np.asarray(entry.get("d", None)),
                "width": entry.get("image_width", None), "height": entry.get("image_height", None),
                "hfov": entry.get("fov", None)}

    def get_odometry(self, name_txt):
        return np.loadtxt(os.path.join(self.pack_path, "odometry", name_txt))

    # ------------------------------------------------------------------ per-sensor iterators
    @staticmethod
    def _stamp_of(path):
        """<dir>/<timestamp_ms>.<ext> -> seconds (dataset_reader.py:94-99)."""
        return float(os.path.basename(path).split(".")[0]) / 1000.0

    def _yield_json_frames(self, sensor_dir, stamps_ms):
        """(seconds, frame dict) for every stamp of a sensor logged into one <pack>/<sensor_dir>/data.json."""
        with open(os.path.join(self.pack_path, sensor_dir, "data.json"), "r") as f:
            frames = json.load(f)
        for ts in stamps_ms:
            yield float(ts) / 1000.0, frames[str(ts)]

    def yield_lidar(self, start_idx=None, end_idx=None, deskewed=False):
        """(seconds, (n,6) float64 [x y z intensity ring t]) per sweep (dataset_reader.py:45-51)."""
        for path in self.yield_sensor_filepath("lidar_top", "bin", start_idx=start_idx, end_idx=end_idx):
            if deskewed:
                path = path.replace("lidar_top", "deskewed_lidar_top")
            yield self._stamp_of(path), np.fromfile(path, dtype=np.double).reshape(-1, 6)

    def yield_IMU(self, start_idx=None, end_idx=None, start_time=None, end_time=None):
        """All unsynced IMU frames; the range arguments are accepted and ignored like the reference's (:53-61)."""
        yield from self._yield_json_frames("IMU", self.attribute["unsync"]["IMU"])

    def yield_GNSS(self, start_idx=None, end_idx=None):
        yield from self._yield_json_frames("UB482", self.attribute["unsync"]["UB482"])       # :63-70

    def yield_wheel(self, sync=True, start_idx=None, end_idx=None):
        yield from self._yield_json_frames("wheel", self.attribute["sync" if sync else "unsync"]["wheel"])   # :85-92

    def yield_camera(self, camera="camera_front", start_idx=None, end_idx=None):
        """(seconds, (H,W,3) uint8 BGR) per frame (dataset_reader.py:72-76)."""
        from .frames import read_bgr
        for path in self.yield_sensor_filepath(camera, "jpg", start_idx=start_idx, end_idx=end_idx):
            yield self._stamp_of(path), read_bgr(path)

    def yield_semantic(self, camera="camera_front", start_idx=None, end_idx=None):
        """(seconds, label image as stored: (H,W) or (H,W,3|4) in BGR(A) order) from <pack>/seg_<camera>/ (:78-83)."""
        from PIL import Image
        for path in self.yield_sensor_filepath(camera, "png", start_idx=start_idx, end_idx=end_idx):
            path = path.replace(camera, "seg_" + camera)
            with Image.open(path) as im:
                arr = np.array(im.convert("RGB") if im.mode == "P" else im)
            if arr.ndim == 3 and arr.shape[2] >= 3:
                arr = np.ascontiguousarray(arr[..., [2, 1, 0] + list(range(3, arr.shape[2]))])
            yield self._stamp_of(path), arr

    # ------------------------------------------------------------------ GNSS / wheel odometry as TUM rows
    def get_GNSS_tum(self):
        """(n,8) [t x y z qx qy qz qw] from the GNSS log; "position" is a dict (x,y,z / x,y,z,w) in current logs and
        a list in the deprecated packstreamer format, decided on the first frame (dataset_reader.py:296-348)."""
        rows, keyed = [], None
        for t, g in self.yield_GNSS():
            if keyed is None:
                keyed = "x" in g["position"]
            if keyed:
                p, q = g["position"], g["orientation"]
                rows.append([t, p["x"], p["y"], p["z"], q["x"], q["y"], q["z"], q["w"]])
            else:
                _warn_packstreamer()
                rows.append([t, *g["position"][:3], *g["orientation"][:4]])
        return np.asarray(rows)

    def get_wheel_tum(self, sync=False):
        """(n,8) TUM rows from wheel odometry: full roll/pitch/yaw + z in the deprecated format, planar (yaw only,
        z = 0) in the current one; quaternion from intrinsic XYZ euler angles (dataset_reader.py:350-407)."""
        from scipy.spatial.transform import Rotation
        frames = list(self.yield_wheel(sync=sync))
        if not frames:
            return np.asarray([])
        full = "roll" in frames[0][1]
        if full:
            for _ in frames:
                _warn_packstreamer()
            rpy = np.array([[w["roll"], w["pitch"], w["yaw"]] for _, w in frames], dtype=np.float64)
            z = [w["z"] for _, w in frames]
        else:
            rpy = np.array([[0, 0, w["yaw"]] for _, w in frames], dtype=np.float64)
            z = [0] * len(frames)
        quat = Rotation.from_euler("XYZ", rpy, degrees=False).as_quat()
        return np.asarray([[t, w["x"], w["y"], z[i], *quat[i]] for i, (t, w) in enumerate(frames)])


def _warn_packstreamer():
    from warnings import warn
    warn("Warning(Deprecation): clip/pack results extracted by packstreamer will not be supported in the future")
