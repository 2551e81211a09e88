"""MapManager / CameraManager with the reference's public surface (cama/reproject.py), backed by the
HIP kernels in libcama_hip.so.

What runs where
  * per clip, host (numpy, vectorised; bit-identical to the reference's Python loops, pinned by
    tests/golden): label densification, BEV-height lookup, pixel->world (reproject.py:36-106), K scaling
    (reproject.py:176-182).
  * per frame, device (C ABI): homogeneous transforms, crop test, K / divide / cull, disc stamping
    (reproject.py:108-131,187-205,246-257).  The list-of-dict methods below upload their arguments, run the
    kernels and download the result, so they are drop-in but pay PCIe both ways; ClipManager's lazy frame
    objects keep everything on the device instead (cama_amd/dataset.py).
There is no CPU fallback for the device parts: without libcama_hip.so or a GPU they raise.
"""
from os.path import join

import numpy as np

from .dataset_reader import DatasetReader
from . import runtime


def colour_id_of(class_name):
    """reproject.py:250-253: lane_marking keeps its own colour, every other class is drawn as Crosswalk_Line."""
    return 0 if class_name == "lane_marking" else 1


def flatten_instances(instances, width=3):
    """list[{"class","points"}] -> (points (n,width), counts (I,), classes list)."""
    classes = [ins["class"] for ins in instances]
    counts = np.asarray([np.asarray(ins["points"]).shape[0] for ins in instances], np.int64)
    if len(instances):
        pts = np.concatenate([np.asarray(ins["points"]).reshape(-1, width) for ins in instances], axis=0)
    else:
        pts = np.zeros((0, width))
    return pts, counts, classes


def split_instances(points, counts, classes, keep=None, joined=False):
    """Inverse of flatten_instances; `keep` (n,) bool drops points and then empty instances.  joined=True (the opt-in
    segment extension only) adds a key "joined" (bool per surviving point): the point's predecessor in the instance
    survived too, i.e. the two are neighbours on the densified polyline and may be connected by a segment."""
    out, o = [], 0
    for cls, n in zip(classes, counts):
        p = points[o:o + n]
        j = None
        if keep is not None:
            k = keep[o:o + n]
            if joined:
                prev = np.concatenate([[False], k[:-1]])
                j = prev[k]
            p = p[k]
            if p.shape[0] == 0:
                o += n
                continue
        elif joined:
            j = np.arange(n) > 0
        ins = {"class": cls, "points": p}
        if joined:
            ins["joined"] = j
        out.append(ins)
        o += n
    return out


class BaseManager:
    def __init__(self):
        pass

    @staticmethod
    def get_color_maps():
        # RGB; reproject.py:11-17
        return {"Road_teeth": np.array([235, 73, 127]),
                "lane_marking": np.array([211, 211, 211]),
                "Stop_Line": np.array([211, 211, 211]),
                "Crosswalk_Line": np.array([255, 215, 0])}


class MapManager(BaseManager):
    def __init__(self):
        super().__init__()
        self.solution = 0.1      # label units per densified point, and metres per BEV pixel
        self.center_x = 0
        self.center_y = 0
        self.map_width = 600
        self.map_height = 600
        self.crop_dict = {"x_min": -50, "x_max": 50, "y_min": -100, "y_max": 100, "z_min": -200, "z_max": 200}

    # ------------------------------------------------------------------ per-clip static map (host)
    def pixel2world_xy(self, pixel_xy):
        """BEV pixel (px, py) -> world (x, y) with the axis swap of reproject.py:36-40."""
        world = np.zeros_like(pixel_xy)
        world[:, 0] = pixel_xy[:, 1] * self.solution - self.map_width / 2 + self.center_x
        world[:, 1] = pixel_xy[:, 0] * self.solution - self.map_height / 2 + self.center_y
        return world

    def _densify(self, line_points):
        """(k,2) float32 polyline -> densified float32 points: segment s contributes
        start + (end-start)/num*j, j = 0..num-1, num = int(|seg| / solution); the end point is never emitted
        and segments with num == 0 vanish (reproject.py:51-63).  Vectorised, same float32 operations."""
        delta = line_points[1:] - line_points[:-1]
        num = (np.linalg.norm(delta, axis=-1) / self.solution).astype(np.int64)
        total = int(num.sum())
        if total == 0:
            # the reference indexes an empty 1-D array here (reproject.py:65 / :96-97)
            raise IndexError("too many indices for array: label densifies to zero points")
        seg = np.repeat(np.arange(num.shape[0]), num)
        first = np.cumsum(num) - num
        j = (np.arange(total) - np.repeat(first, num)).astype(np.float32)
        step = delta[seg] / num[seg].astype(np.float32)[:, None]
        return line_points[:-1][seg] + step * j[:, None]

    def segment_table(self, maps_2d):
        """Host half of the device static-map build: the O(#label vertices) part of reproject.py:44-63 / :74-93.
        Returns dict(verts (V,2) f32, seg_v0, seg_num (S,) i32, seg_off (S+1,) i64, seg_colour (S,) u8,
        counts (I,) i64 points per kept label, classes [I]).  Same float32 operations as _densify; a label whose
        segments all round to zero points raises IndexError like the reference."""
        verts, seg_v0, seg_num, seg_colour, counts, classes = [], [], [], [], [], []
        v_base = 0
        for label in maps_2d:
            pts = label["data"]
            if len(pts) <= 1:
                continue
            line = np.array(pts).astype(np.float32)
            delta = line[1:] - line[:-1]
            num = (np.linalg.norm(delta, axis=-1) / self.solution).astype(np.int64)
            total = int(num.sum())
            if total == 0:
                raise IndexError("too many indices for array: label densifies to zero points")
            keep = np.flatnonzero(num > 0)
            verts.append(line)
            seg_v0.append(v_base + keep)
            seg_num.append(num[keep])
            seg_colour.append(np.full(keep.shape[0], colour_id_of(label["attrs"]["type"]), np.uint8))
            counts.append(total)
            classes.append(label["attrs"]["type"])
            v_base += line.shape[0]
        if not counts:
            return {"verts": np.zeros((0, 2), np.float32), "seg_v0": np.zeros(0, np.int32),
                    "seg_num": np.zeros(0, np.int32), "seg_off": np.zeros(1, np.int64),
                    "seg_colour": np.zeros(0, np.uint8), "counts": np.zeros(0, np.int64), "classes": []}
        num_all = np.concatenate(seg_num)
        return {"verts": np.ascontiguousarray(np.concatenate(verts)),
                "seg_v0": np.concatenate(seg_v0).astype(np.int32), "seg_num": num_all.astype(np.int32),
                "seg_off": np.concatenate([[0], np.cumsum(num_all)]).astype(np.int64),
                "seg_colour": np.concatenate(seg_colour), "counts": np.asarray(counts, np.int64), "classes": classes}

    def load_3d_instance_maps(self, maps_2d):
        """nuScenes labels (metres, z = 0): reproject.py:42-70."""
        instances = []
        for label in maps_2d:
            pts = label["data"]
            if len(pts) <= 1:
                continue
            dense = self._densify(np.array(pts).astype(np.float32))
            z = np.zeros_like(dense[:, 0])
            instances.append({"class": label["attrs"]["type"],
                              "points": np.concatenate((dense, z[:, None]), axis=-1).reshape(-1, 3)})
        return instances

    def calculate_3d_instance_maps(self, bev_height, maps_2d):
        """CAMA labels (BEV pixels) lifted with the height raster: reproject.py:72-106."""
        instances = []
        for label in maps_2d:
            pts = label["data"]
            if len(pts) <= 1:
                continue
            dense = self._densify(np.array(pts).astype(np.float32))
            pix = dense.round().astype(np.uint16)[:, ::-1].clip(0, bev_height.shape[0] - 1)   # (row, col)
            h = bev_height[pix[:, 0], pix[:, 1]]
            xy = self.pixel2world_xy(dense)
            instances.append({"class": label["attrs"]["type"],
                              "points": np.concatenate((xy, h[:, None]), axis=-1).reshape(-1, 3)})
        return instances

    # ------------------------------------------------------------------ per-frame geometry (device)
    def transform_3d_instance_maps(self, maps, transform):
        """Homogeneous 4x4 on every instance (reproject.py:108-116); float64 result."""
        maps = list(maps)
        pts, counts, classes = flatten_instances(maps)
        if pts.shape[0] == 0:
            return [{"class": c, "points": np.zeros((0, 3))} for c in classes]
        out, _ = runtime.engine().transform_points(pts, np.asarray(transform, np.float64)[None])
        return split_instances(out[0].cpu().numpy(), counts, classes)

    def crop_3d_instance_maps(self, maps, crop_dict=None):
        """Inclusive axis-aligned box; empty instances are dropped (reproject.py:118-131)."""
        box = crop_dict if crop_dict is not None else self.crop_dict
        maps = list(maps)
        pts, counts, classes = flatten_instances(maps)
        if pts.shape[0] == 0:
            return []
        crop = [box["x_min"], box["x_max"], box["y_min"], box["y_max"], box["z_min"], box["z_max"]]
        keep = runtime.engine().crop_points(pts, crop).cpu().numpy().astype(bool)
        return split_instances(pts, counts, classes, keep)

    def crop_box(self):
        b = self.crop_dict
        return [b["x_min"], b["x_max"], b["y_min"], b["y_max"], b["z_min"], b["z_max"]]

    # ------------------------------------------------------------------ debug dumps (unused by the demo)
    def save_pcd(self, maps, pcd_path):
        import open3d as o3d
        pts, counts, classes = flatten_instances(maps)
        colours = np.concatenate([np.tile(self.get_color_maps()[c], (n, 1)) for c, n in zip(classes, counts)], axis=0)
        pcd = o3d.geometry.PointCloud()
        pcd.points = o3d.utility.Vector3dVector(pts)
        pcd.colors = o3d.utility.Vector3dVector(colours / 255.)
        o3d.io.write_point_cloud(pcd_path, pcd)

    def save_xyz(self, maps, xyz_path):
        np.savetxt(xyz_path, flatten_instances(maps)[0], fmt="%.3f")


class CameraManager(BaseManager):
    def __init__(self, clip_path, camera_name, output_size=(540, 960), undisort=True, reader=None):
        super().__init__()
        # `reader` (extension): a DatasetReader of this clip that the caller already has -- ClipManager parses
        # attribute.json ONCE for its six cameras instead of once per camera like the reference (reproject.py:166)
        dr = reader if reader is not None else DatasetReader(clip_path)
        self.dr = dr
        self.clip_path = clip_path
        self.camera_name = camera_name
        self.chassis2camera = dr.get_extrinsic("chassis", camera_name)
        intr = dr.get_intrinsics(camera_name)
        self.K_origin = intr["K"]
        self.d_origin = intr["d"]
        self.width_origin = intr["width"]
        self.height_origin = intr["height"]
        self.width = output_size[1]
        self.height = output_size[0]
        if undisort:
            self.d = []
        # K follows the output size: row 0 scales with width, row 1 with height (reproject.py:180-182)
        self.K = self.K_origin.copy()
        self.K[0, :] = self.K[0, :] * self.width / self.width_origin
        self.K[1, :] = self.K[1, :] * self.height / self.height_origin

    def get_chassis2camera(self):
        return self.chassis2camera

    def project_to_image(self, maps):
        """Camera-frame instances -> (v,u) float64 of the points with z > 0 inside the image; empty instances
        dropped (reproject.py:187-205).  Runs cama_project_points with an identity extrinsic."""
        maps = list(maps)
        pts, counts, classes = flatten_instances(maps)
        if pts.shape[0] == 0:
            return []
        eng = runtime.engine()
        rig = eng.make_rig([self.camera_name], [np.eye(4)], [self.K], self.width, self.height)
        vu, vis = eng.project_points(rig, pts)
        return split_instances(vu[0].cpu().numpy(), counts, classes, vis[0].cpu().numpy().astype(bool))

    # ------------------------------------------------------------------ frame files
    def index2timestamp(self, index, sync):
        return self.dr.attribute["sync" if sync else "unsync"][self.camera_name][index]

    def get_image_path(self, index, sync):
        return join(self.clip_path, self.camera_name, f"{self.index2timestamp(index, sync)}.jpg")

    def get_instance_path(self, index, sync=True):
        return join(self.clip_path, f"lane_ins_{self.camera_name}", f"{self.index2timestamp(index, sync)}.png")

    def read_resized_instance_by_index(self, index, sync=True):
        raise NotImplementedError("instance-mask frames are not part of the reprojection demo (reproject.py:222-226 "
                                  "is unused by main.py)")

    def read_resized_image_by_index(self, index, sync=True):
        return self.read_resized_image(self.get_image_path(index, sync))

    def needs_resample(self):
        d = self.d_origin if self.d == [] else self.d
        distorted = d is not None and np.size(d) > 0 and bool(np.any(np.asarray(d, np.float64) != 0))
        return distorted or (self.width, self.height) != (self.width_origin, self.height_origin)

    def resize_image(self, image, interpolation=None):
        """Undistort + resize to (height, width) (reproject.py:232-240).  Identity when the output size equals the
        source size and the distortion is zero; otherwise the frame goes through the device resampler."""
        if not self.needs_resample():
            return np.ascontiguousarray(image)
        from . import frames
        return frames.resample_host_image(self, image)

    def read_resized_image(self, image_path):
        from . import frames
        return self.resize_image(frames.read_bgr(image_path))

    # ------------------------------------------------------------------ raster
    def render_maps(self, image, maps_2d, segments=False):
        """Draw every point as a filled radius-2 disc, in order, later discs on top (reproject.py:246-257).
        `image` (H,W,3) uint8 BGR is updated in place and returned, like cv2.circle does.
        segments=True: EXTENSION without reference semantics (the reference draws discs only; BASELINE.json's north_star
        asks for rasterised segments) -- neighbouring points of an instance are also joined by one-pixel Bresenham segments
        (an instance's "joined" flags when present, else every point to its predecessor); cama_stamp_polylines.
        segments="wu": anti-aliased Wu lines instead, blended once per pixel by coverage; cama_stamp_polylines_wu."""
        import torch
        wu = isinstance(segments, str) and segments.lower() == "wu"      # anti-aliased variant: cama_stamp_polylines_wu
        maps_2d = list(maps_2d)
        vu, counts, classes = flatten_instances(maps_2d, width=2)
        if vu.shape[0] == 0:
            return image
        colour = np.repeat(np.asarray([colour_id_of(c) for c in classes], np.uint8), counts)
        link = None
        if segments:
            link = np.concatenate([np.asarray(ins["joined"], bool) if "joined" in ins else np.arange(len(ins["points"])) > 0
                                   for ins in maps_2d if len(ins["points"])])
        eng = runtime.engine()
        dev = torch.from_numpy(np.ascontiguousarray(image)).to(eng.device)
        eng.stamp_points(dev, vu, colour, link=link, wu=wu)
        image[...] = dev.cpu().numpy()
        return image
