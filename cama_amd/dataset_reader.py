"""Clip ("pack") reader: the subset of the reference's DatasetReader (cama/dataset_reader.py) that the
reprojection path touches -- attribute.json, sensor timestamps, the calibration graph, intrinsics,
odometry text files, sensor file paths.  Pure host-side file parsing, once per clip.

Out of scope: the lidar / IMU / GNSS / wheel iterators and their TUM converters
(dataset_reader.py:45-71,86-93,296-407); they raise NotImplementedError here.
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

    # ------------------------------------------------------------------ out of scope
    def _out_of_scope(self, *a, **k):
        raise NotImplementedError("sensor iterators other than camera paths are outside the reprojection path "
                                  "(SURVEY.md section 8); use the reference's DatasetReader for them")

    yield_lidar = yield_IMU = yield_GNSS = yield_wheel = yield_camera = yield_semantic = _out_of_scope
    get_GNSS_tum = get_wheel_tum = _out_of_scope
