"""ClipManager with the reference's surface (cama/dataset.py), running the per-frame work on MI355X.

main.py's loop is unchanged:

    cm = ClipManager(configs["cama_configs"], clip_path)
    for image_idx, instance_map in cm.yield_frame(dataset="cama"):
        maps_2d_dict = cm.project_all_camera(instance_map)
        image_dict = cm.render_vectors(maps_2d_dict, image_idx)
        image = vg.concate_image(image_dict)

but the three objects that flow through it are lazy handles (FrameMaps -> ProjectedMaps -> RenderedFrame):
as long as the caller only passes them along, nothing is materialised on the host and each frame is one
fused device pass (cama_render_frames).  Indexing / iterating a handle materialises exactly what the
reference would have returned (list of {"class","points"} dicts, float64, same order, empty instances
dropped), computed by the same kernels in their coordinate-emitting mode.  Plain lists / dicts of numpy
arrays are accepted too and take the generic upload -> kernel -> download path.

Extensions beyond the reference surface (used by bench.py and by batch pipelines):
    frame_poses(dataset)            all frames' world->chassis float32 matrices in one vectorised pass
    render_clip(dataset, ...)       whole-clip fused render, F frames per launch, output stays in HBM
    set_frame_source(source)        where camera frames come from (disk by default, HBM tensors for bench)
Reference lines: cama/dataset.py:11-126.
"""
import os
from collections.abc import Mapping, Sequence
from os.path import exists, join

import numpy as np

from . import runtime
from .dataset_reader import DatasetReader
from .engine import ChunkedMosaic
from .pose_transformer import PoseTransformer
from .reproject import CameraManager, MapManager, colour_id_of, flatten_instances, split_instances
from .tools import load_json

# Module switches (A/B and test hooks; speed only -- none can change a byte).  They were environment variables until round 5.
POSE_MEMO = True                     # frame_poses() memoised per (track, stamps); False: seek + slerp + inverse on every call
LAUNCH_MEMO = True                   # a pipelined render_clip into the same buffers replays its launch list (one library call each)
MOSAIC_CHUNK_BYTES = 8 << 30         # clips beyond this get a ChunkedMosaic (one pooled buffer per launch) from render_clip(out=None)

try:                                    # progress bar exactly like the reference when tqdm is there
    from tqdm import tqdm as _tqdm
except ImportError:                     # pragma: no cover
    def _tqdm(it, **_):
        return it


class _StaticMap:
    """One dataset pass' static instances + their flat device copy (uploaded on first GPU use)."""

    def __init__(self, instances):
        self.instances = instances
        self._dmap = None
        self._xyz = None
        if isinstance(instances, StaticInstances):            # built on the device: no host copy unless asked for
            self._dmap = instances.dmap
            self.counts, self.classes = instances.counts, instances.classes
        else:
            self._xyz, self.counts, self.classes = flatten_instances(instances)
            if self._xyz.dtype not in (np.float32, np.float64):
                self._xyz = self._xyz.astype(np.float64)
            self.colour = np.repeat(np.asarray([colour_id_of(c) for c in self.classes], np.uint8), self.counts)
            # bit 1: "joined to the previous vertex" = every point but the first of its instance (read by the opt-in
            # segment extension only; the disc path looks at bit 0)
            if self.colour.size:
                first = np.zeros(self.colour.size, bool)
                first[np.concatenate([[0], np.cumsum(self.counts)[:-1]])[np.asarray(self.counts) > 0]] = True
                self.colour = self.colour | (np.uint8(2) * (~first).astype(np.uint8))

    @property
    def xyz(self):
        if self._xyz is None:
            self._xyz, _, _ = flatten_instances(self.instances._materialise())
        return self._xyz

    @property
    def n_points(self):
        return int(self.counts.sum())

    def device(self):
        if self._dmap is None:
            # content-keyed: the scenes of one site (identical site-aggregated labels) share one device map per GPU
            self._dmap = runtime.engine().shared_map(self._xyz, self.colour)
        return self._dmap


class StaticInstances(Sequence):
    """A dataset pass' static map built ON THE DEVICE (configs["device_map_build"]): behaves as the reference's
    list of {"class","points"} dicts (cama/dataset.py:13-24) when touched -- the points are downloaded once --
    while the fused render uses the device buffer directly."""

    def __init__(self, dmap, counts, classes):
        self.dmap, self.counts, self.classes = dmap, np.asarray(counts, np.int64), list(classes)
        self._items = None

    def _materialise(self):
        if self._items is None:
            xyz = np.ascontiguousarray(self.dmap.soa.cpu().numpy().T)
            self._items = split_instances(xyz, self.counts, self.classes)
        return self._items

    def __len__(self):
        return len(self.classes)

    def __getitem__(self, k):
        return self._materialise()[k]


class FrameMaps(Sequence):
    """What yield_frame yields as `instance_map`: the clip's instances in the chassis frame of one image,
    cropped (cama/dataset.py:99-106).  Lazy; behaves as the reference's list of dicts when touched."""

    def __init__(self, owner, dataset, image_idx, world2chassis):
        self.owner, self.dataset, self.image_idx = owner, dataset, image_idx
        self.world2chassis = world2chassis          # (4,4) float32, the np.linalg.inv result
        self._items = None

    def _materialise(self):
        if self._items is None:
            sm = self.owner._static(self.dataset)
            eng = runtime.engine()
            if sm.n_points == 0:
                self._items = []
            else:
                out, mask = eng.transform_points(sm.xyz, self.world2chassis[None], crop=self.owner.mm.crop_box())
                self._items = split_instances(out[0].cpu().numpy(), sm.counts, sm.classes,
                                              mask[0].cpu().numpy().astype(bool))
        return self._items

    def __len__(self):
        return len(self._materialise())

    def __getitem__(self, k):
        return self._materialise()[k]


class ProjectedMaps(Mapping):
    """What project_all_camera returns for a FrameMaps: camera name -> list of {"class","points": (m,2) (v,u)}
    (cama/dataset.py:108-117).  Lazy."""

    def __init__(self, frame):
        self.frame = frame
        self._items = None

    def _materialise(self):
        if self._items is None:
            fr = self.frame
            owner = fr.owner
            sm = owner._static(fr.dataset)
            names = [cm.camera_name for cm in owner.cm_list]
            if sm.n_points == 0:
                self._items = {n: [] for n in names}
            else:
                eng = runtime.engine()
                vu, vis, _ = eng.project_frames(sm.device(), owner._rig(), fr.world2chassis[None],
                                                crop=owner.mm.crop_box())
                vu, vis = vu[0].cpu().numpy(), vis[0].cpu().numpy().astype(bool)
                seg = bool(owner.configs.get("segments", False))      # (opt-in extension: keep who neighbours whom)
                self._items = {n: split_instances(vu[c], sm.counts, sm.classes, vis[c], joined=seg)
                               for c, n in enumerate(names)}
        return self._items

    def __getitem__(self, name):
        return self._materialise()[name]

    def __iter__(self):
        return iter([cm.camera_name for cm in self.frame.owner.cm_list])

    def __len__(self):
        return len(self.frame.owner.cm_list)


class RenderedFrame(Mapping):
    """What render_vectors returns on the fused path: camera name -> (H,W,3) uint8 BGR ndarray
    (cama/dataset.py:119-126), backed by the mosaic the overlay kernel wrote in HBM.  `batch` / `j`: the render-ahead
    batch the frame belongs to (cama_amd/egress.py), which VideoGenerator uses for the device-side egress."""

    def __init__(self, names, mosaic_dev, H, W, cols=3, batch=None, j=0):
        self.names, self.mosaic_device, self.H, self.W, self.cols = list(names), mosaic_dev, H, W, cols
        self.batch, self.j = batch, j
        self._host = None

    def _mosaic_host(self):
        if self._host is None:
            # (the batch's pinned host copy when a bgr24 download was started behind the render, else one download)
            self._host = self.batch.bgr(self.j) if self.batch is not None else self.mosaic_device.cpu().numpy()
        return self._host

    def mosaic(self, order=None):
        """The 2x3 mosaic as ndarray if the camera order matches the kernel's cell layout, else None."""
        if order is not None and list(order) != self.names:
            return None
        return self._mosaic_host()

    def mosaic_handle(self, order=None):
        """The mosaic without leaving HBM (egress.DeviceMosaic) when the frame came out of a render batch and the
        camera order matches the kernel's cell layout, else None."""
        if self.batch is None or (order is not None and list(order) != self.names):
            return None
        from .egress import DeviceMosaic
        return DeviceMosaic(self.batch, self.j)

    def __getitem__(self, name):
        c = self.names.index(name)
        r, q = divmod(c, self.cols)
        return self._mosaic_host()[r * self.H:(r + 1) * self.H, q * self.W:(q + 1) * self.W]

    def __iter__(self):
        return iter(self.names)

    def __len__(self):
        return len(self.names)


def _segments_mode(v):
    """configs["segments"] / segments= argument -> False, True (one-pixel Bresenham segments) or "wu" (anti-aliased)."""
    if isinstance(v, str):
        if v.lower() == "wu":
            return "wu"
        if v.lower() in ("bresenham", "true", "1"):
            return True
        if v.lower() in ("", "false", "0", "none"):
            return False
        raise ValueError(f"segments = {v!r}: expected False, True, 'bresenham' or 'wu'")
    return bool(v)


class ClipManager:
    def __init__(self, configs, clip_path=None, output_size=None):
        self.configs = configs
        self.mm = MapManager()
        self.instance_maps = dict()
        self.output_size = tuple(output_size) if output_size is not None else \
            tuple(configs.get("output_size", (540, 960)))
        self._static_cache = {}
        self._track_cache = {}
        self._poses_memo = {}
        self._launch_memo = {}
        self._rig_cache = None
        # configs["egress"] = "i420" | "bgr24": what a listening VideoGenerator receives (runtime.egress_format); absent =
        # the environment's choice (default bgr24, the reference's stream).  Process-wide: the newest ClipManager decides.
        runtime.set_egress_format(configs.get("egress") if isinstance(configs, dict) else None)
        self._frame_source = None
        if clip_path is not None:
            self.clip_path = clip_path
            self.cm_list = self.prepare_camera_manager(clip_path)
            cama_instance = self.load_clip_cama(clip_path)
            if cama_instance is not None:
                self.instance_maps["cama"] = cama_instance
            nuscenes_instance = self.load_clip_nuscenes(clip_path)
            if nuscenes_instance is not None:
                self.instance_maps["nuscenes"] = nuscenes_instance

    # ------------------------------------------------------------------ per-clip setup (host)
    def load_clip_cama(self, clip_path):
        label_json = join(clip_path, self.configs["result_dir"], self.configs["cama_map_file"])
        if not exists(label_json):
            return None
        raster = join(clip_path, self.configs["result_dir"], self.configs["height_mlp"])
        labels = load_json(label_json)
        if self.configs.get("device_map_build", False):
            return self._build_on_device(labels, np.load(raster))
        # host build: the raster is only GATHERED at the densified points (cama/reproject.py:96-99) -- map the file instead
        # of reading all of it (6000 x 6000 float32 = 144 MB per clip in the reference's data; same values either way)
        try:
            bev_height = np.load(raster, mmap_mode="r")
        except ValueError:                          # (pickled / non-mappable arrays)
            bev_height = np.load(raster)
        return self.mm.calculate_3d_instance_maps(bev_height, labels)

    def load_clip_nuscenes(self, clip_path):
        label_json = join(clip_path, self.configs["result_dir"], self.configs["nuscenes_map_file"])
        if not exists(label_json):
            return None
        labels = load_json(label_json)
        if self.configs.get("device_map_build", False):
            return self._build_on_device(labels, None)
        return self.mm.load_3d_instance_maps(labels)

    def _build_on_device(self, labels, bev_height):
        """Static map straight into HBM (cama_build_static_map); bit-identical to MapManager's host build."""
        mm = self.mm
        table = mm.segment_table(labels)
        dmap = runtime.engine().build_static_map(table, lift=bev_height is not None, bev_height=bev_height,
                                                 solution=mm.solution, map_width=mm.map_width, map_height=mm.map_height,
                                                 center_x=mm.center_x, center_y=mm.center_y)
        if self.configs.get("segments", False) and dmap.N:
            # the segment extension reads "joined to the previous vertex" from bit 1 of the colour byte: every point but
            # the first of its instance
            import torch
            counts = np.asarray(table["counts"], np.int64)
            starts = np.concatenate([[0], np.cumsum(counts)[:-1]])[counts > 0]
            dmap.colour |= 2
            dmap.colour[torch.from_numpy(starts).to(dmap.colour.device)] &= 1
            dmap.has_links = True
        return StaticInstances(dmap, table["counts"], table["classes"])

    def prepare_camera_manager(self, clip_path):
        # one parse of attribute.json for the clip (six cameras + the pose tracks read the same dict; read-only)
        self._reader = DatasetReader(clip_path)
        return [CameraManager(clip_path, name, output_size=self.output_size, reader=self._reader)
                for name in self.configs["camera_list"]]

    def get_pt_cama(self, dr):
        """camera_main -> world track from the label zip, right-multiplied by chassis -> camera_main:
        chassis -> world (cama/dataset.py:60-69)."""
        main = self.configs["camera_main"]
        pt = PoseTransformer()
        pt.loadarray(dr.get_odometry(f"{self.configs['pose_prefix']}_{main}.txt"))
        pt.right_rotate(dr.get_extrinsic("chassis", main))
        return pt

    def get_pt_nuscenes(self, dr):
        """chassis -> (offset) world track re-expressed in its middle pose (cama/dataset.py:71-76)."""
        pt = PoseTransformer()
        pt.loadarray(dr.get_odometry("wigo_offset_clip.txt"))
        pt.normalize2center()
        return pt

    # ------------------------------------------------------------------ device-side state
    def _static(self, dataset):
        ins = self.instance_maps[dataset]
        sm = self._static_cache.get(dataset)
        if sm is None or sm.instances is not ins:
            sm = _StaticMap(ins)
            self._static_cache[dataset] = sm
        return sm

    def _rig(self):
        if self._rig_cache is None:
            cms = self.cm_list
            self._rig_cache = runtime.engine().make_rig(
                [c.camera_name for c in cms], [c.get_chassis2camera() for c in cms], [c.K for c in cms],
                cms[0].width, cms[0].height)
        return self._rig_cache

    def set_frame_source(self, source):
        self._frame_source = source

    def frame_source(self):
        if self._frame_source is None:
            from .frames import ClipFrameSource
            self._frame_source = ClipFrameSource(self.cm_list, runtime.engine().device)
        return self._frame_source

    # ------------------------------------------------------------------ poses
    def _track(self, dataset):
        """(PoseTransformer, frame seconds) of one dataset pass; parsed from the clip files once and kept, like the
        static maps (the reference re-reads them on every yield_frame call, cama/dataset.py:80-87)."""
        hit = self._track_cache.get(dataset)
        if hit is None:
            dr = getattr(self, "_reader", None) or DatasetReader(self.clip_path)
            if dataset == "nuscenes":
                pt = self.get_pt_nuscenes(dr)
            elif dataset == "cama":
                pt = self.get_pt_cama(dr)
            else:
                raise UnboundLocalError(f"unknown dataset {dataset!r}")   # the reference fails on `pt` here
            hit = (pt, dr.get_sensor_timestamp(self.configs["camera_main"], sync=True))
            self._track_cache[dataset] = hit
        return hit

    def frame_poses(self, dataset):
        """(image indices (F,), world->chassis (F,4,4) float32) for every renderable frame of the clip.

        Same arithmetic as the per-frame body of yield_frame (cama/dataset.py:87-99) -- interpolated float64
        chassis->world, cast to float32, float32 general inverse -- done for all frames at once.  Frames whose
        pose lookup would raise RuntimeError are left out, as the reference skips them.  Index 0 is never
        rendered (dataset.py:88 starts at 1)."""
        track = self._track(dataset)
        # memoised per (track, stamps): both are parsed from the clip's files once and never change afterwards, so neither does
        # this function of them -- the same arrays come back (read-only), bit for bit what the recomputation gives
        # (tests/test_host_golden.py).  On launches of 0.1 ms the 0.1 ms of seek + slerp + inverse per step was what paced the
        # host (profiles/r05_960x540_bench.json).  POSE_MEMO = False (module switch): recompute on every call (A/B).
        memo = self._poses_memo.get(dataset)
        if memo is not None and memo[0] is track and POSE_MEMO:
            return memo[1]
        pt, secs = track
        if len(secs) <= 1:
            return np.zeros(0, np.int64), np.zeros((0, 4, 4), np.float32)
        ok, c2w = pt.seek_many(secs[1:], 0.5, interpolate=True)
        idx = np.flatnonzero(ok) + 1
        c2w32 = c2w[ok].astype(np.float32)
        w2c = np.ascontiguousarray(np.linalg.inv(c2w32)) if len(idx) else np.zeros((0, 4, 4), np.float32)
        idx.setflags(write=False)
        w2c.setflags(write=False)
        self._poses_memo[dataset] = (track, (idx, w2c))
        return idx, w2c

    # ------------------------------------------------------------------ the reference's per-frame API
    def yield_frame(self, dataset):
        idx, w2c = self.frame_poses(dataset)
        lookup = {int(i): k for k, i in enumerate(idx)}
        n_stamps = len(self._track(dataset)[1])
        for image_idx in _tqdm(range(1, n_stamps)):
            k = lookup.get(image_idx)
            if k is None:
                continue            # pose lookup failed: frame skipped (cama/dataset.py:93-96)
            yield (image_idx, FrameMaps(self, dataset, image_idx, w2c[k]))

    def project_all_camera(self, maps_3d):
        if isinstance(maps_3d, FrameMaps) and maps_3d.owner is self and maps_3d._items is None:
            return ProjectedMaps(maps_3d)
        # generic path: caller-supplied chassis-frame instances
        pts, counts, classes = flatten_instances(list(maps_3d))
        names = [cm.camera_name for cm in self.cm_list]
        if pts.shape[0] == 0:
            return {n: [] for n in names}
        vu, vis = runtime.engine().project_points(self._rig(), pts)
        vu, vis = vu.cpu().numpy(), vis.cpu().numpy().astype(bool)
        return {n: split_instances(vu[c], counts, classes, vis[c]) for c, n in enumerate(names)}

    def render_vectors(self, maps_2d_dict, image_idx):
        segments = _segments_mode(self.configs.get("segments", False))
        # configs["segments"] = True | "wu" (EXTENSION, no reference semantics): discs + one-pixel segments between neighbouring
        # points; caller-supplied 2D instances go image by image through the generic path (cama_stamp_polylines)
        # (round 4: the fused path draws them too -- CAMA_BIN_SEGMENTS -- as long as the source frames are pre-resized)
        fused_ok = not segments or not (getattr(self.frame_source(), "fused", False)
                                        and hasattr(self.frame_source(), "raw_batch"))
        if fused_ok and isinstance(maps_2d_dict, ProjectedMaps) and maps_2d_dict.frame.owner is self \
                and maps_2d_dict._items is None and maps_2d_dict.frame.image_idx == image_idx:
            fr = maps_2d_dict.frame
            rig = self._rig()
            batch, j = self._render_ahead(fr)
            return RenderedFrame(rig.names, batch.mosaic[j], rig.H, rig.W, batch=batch, j=j)
        # generic path: caller-supplied 2D instances, one image at a time like the reference
        out = {}
        for cm in self.cm_list:
            image = cm.read_resized_image_by_index(image_idx)
            out[cm.camera_name] = cm.render_maps(image, maps_2d_dict[cm.camera_name], segments=segments) if segments \
                else cm.render_maps(image, maps_2d_dict[cm.camera_name])
        return out

    def _render_batch(self, dataset, image_ids, w2c):
        """One fused launch for the frames `image_ids` (consecutive entries of the pass) -> egress.RenderBatch."""
        from .egress import RenderBatch
        eng = runtime.engine()
        rig = self._rig()
        source = self.frame_source()
        dmap = self._static(dataset).device()
        if getattr(source, "fused", False) and hasattr(source, "raw_batch"):
            # raw sensor frames: undistort + resize happens inside the overlay kernel
            mosaic = eng.render_frames_raw(dmap, rig, w2c, source.raw_batch(image_ids), self.cm_list, crop=self.mm.crop_box())
        else:
            mosaic = eng.render_frames(dmap, rig, w2c, source.batch(image_ids), crop=self.mm.crop_box(),
                                       segments=_segments_mode(self.configs.get("segments", False)))
        batch = RenderBatch(eng, image_ids, mosaic)
        fmt = runtime.egress_mode()                     # a VideoGenerator is listening: start the batch's host copy now
        if fmt is not None and not batch.start_egress(fmt) and fmt == "i420":
            batch.start_egress("bgr24")                 # (mosaic shape the I420 converter cannot take)
        return batch

    def _render_ahead(self, fr):
        """(batch, position) holding frame `fr` of a yield_frame pass.  The loop in main.py asks frame by frame, but
        all poses of a pass are known up front (frame_poses), so the frames are rendered `render_ahead` at a time
        (configs["render_ahead"], default 16; 1 = one launch per frame) the first time one of a batch is asked for, and
        the following batch is issued right away so that it runs while this one is consumed.  Same kernels, same
        arguments per frame as a one-frame launch: only the batching differs."""
        B = max(1, int(self.configs.get("render_ahead", 16)))
        # the FIRST batch of a pass is short (configs["render_ahead_first"], default 4): nothing can be handed to the caller
        # before the first batch's files are read, decoded, rendered and downloaded -- with 16 frames that is ~20 ms of a
        # 40-frame scene's pass (profiles/r05_cold_sweep.txt), with 4 a quarter of it; the pump is decoding the next ones by then
        B0 = max(1, min(B, int(self.configs.get("render_ahead_first", 4))))
        crop = tuple(float(v) for v in np.asarray(self.mm.crop_box()).reshape(-1))
        source = self.frame_source()                # create the (lazy) default source BEFORE it goes into the key
        ins = self.instance_maps[fr.dataset]
        key = (fr.dataset, crop, B, B0)
        ra = getattr(self, "_ahead", None)
        # the state holds the objects it was built for (identity, not id(): a held reference cannot be recycled)
        if ra is None or ra["key"] != key or ra["ins"] is not ins or ra["src"] is not source:
            idx, w2c = self.frame_poses(fr.dataset)
            # batches = runs of up to B CONSECUTIVE image indices: a pose gap (skipped frames) ends a batch, so sources
            # that serve contiguous frame ranges only (RawDeviceFrameSource) never see a range with a hole
            bounds, of_pos = [], np.zeros(len(idx), np.int64)
            # batch sizes ramp up: B0, then B / 2 twice, then B -- a pass is a pipeline of decode -> render -> download stages and
            # a nuScenes scene has ~40 frames: with 4 + 16 + 16 + 4 it never fills (cold scene 33.9 ms; with 8s 27.9 ms, same
            # box), while long passes reach the full batch after 4 + 8 + 8 frames
            def cap(n):                                             # size of the n-th batch of the pass (0-based)
                return B0 if n == 0 else (max(B0, B // 2) if n <= 2 else B)
            for k in range(len(idx)):
                if not bounds or k - bounds[-1][0] >= cap(len(bounds) - 1) or idx[k] != idx[k - 1] + 1:
                    bounds.append([k, k])
                bounds[-1][1] = k + 1
                of_pos[k] = len(bounds) - 1
            ra = self._ahead = {"key": key, "ins": ins, "src": source, "idx": idx, "w2c": w2c,
                                "pos": {int(i): k for k, i in enumerate(idx)}, "bounds": bounds, "of_pos": of_pos,
                                "batches": {}}
        k = ra["pos"].get(int(fr.image_idx))
        if k is None or B == 1 or not np.array_equal(ra["w2c"][k], fr.world2chassis):
            return self._render_batch(fr.dataset, [int(fr.image_idx)], np.asarray(fr.world2chassis)[None]), 0
        b = int(ra["of_pos"][k])
        batches = ra["batches"]
        if k < ra.get("last", -1):                  # a new pass over the clip: render it again (files may have changed)
            batches.clear()
        if not batches and hasattr(source, "plan"):
            # every batch of the pass from here on is known: let the source read and decode ahead of the renders
            source.plan([[int(i) for i in ra["idx"][lo:hi]] for lo, hi in ra["bounds"][b:]])
        ra["last"] = k
        # this batch now; the next one a batch early -- but not while the caller still waits for the FIRST frame of this one:
        # issuing a batch means waiting for its frames' decode, and the first frame of a pass used to pay for the second
        # batch's 96 JPEGs as well (profiles/r05_cold_sweep.txt: 12 ms to the first mosaic of a new clip)
        first_of_batch = k == ra["bounds"][b][0]
        for nb in ((b,) if (first_of_batch and b not in batches) else (b, b + 1)):
            if nb not in batches and nb < len(ra["bounds"]):
                lo, hi = ra["bounds"][nb]
                batches[nb] = self._render_batch(fr.dataset, [int(i) for i in ra["idx"][lo:hi]], ra["w2c"][lo:hi])
        for old in [n for n in batches if n < b - 1]:
            del batches[old]
        return batches[b], k - ra["bounds"][b][0]

    # ------------------------------------------------------------------ whole-clip fused path
    def _plan_launches(self, eng, dataset, dmap, rig, src_all, fused_raw, out, w2c, ids, step, cuts, crop, segments, fpl):
        """Work out the pipelined launches of a whole-clip render ONCE: the clip's launch-invariant arguments as a cama_clip
        (Engine.clip_desc) and per launch (host pose pointer, F, source pointer, mosaic pointer, keep tuple).  Kept in
        `_launch_memo[dataset]` together with everything the pointers were taken from; _launches_valid() re-checks those by
        identity before a later render_clip() replays them.  None when this clip does not fit the one-call form."""
        F = len(ids)
        raw = None
        if fused_raw:
            c0 = self.cm_list[0]
            raw = (int(c0.height_origin), int(c0.width_origin), self.cm_list)
        try:
            desc = eng.clip_desc(dmap, rig, crop=crop, segments=segments, raw=raw)
        except Exception:
            return None
        if desc is None:
            return None
        launches, lo = [], 0
        base = w2c.ctypes.data
        while lo < F:
            hi = min([F, lo + step] + [c for c in cuts if c > lo][:1])
            src = src_all.raw_batch(ids[lo:hi]) if fused_raw else src_all.batch(ids[lo:hi])
            dst = out[lo:hi]
            if not (src.is_cuda and src.is_contiguous() and dst.is_contiguous()):
                return None
            want = (hi - lo, rig.C) + ((raw[0], raw[1]) if fused_raw else (rig.H, rig.W)) + (3,)
            assert tuple(src.shape) == want and tuple(dst.shape) == eng.mosaic_shape(rig, hi - lo), (tuple(src.shape), want)
            launches.append((base + lo * 64, hi - lo, src.data_ptr(), dst.data_ptr(), (None, src, dst, dmap, rig, tuple(desc._keep[2:]))))
            lo = hi
        memo = {"desc": desc, "launches": launches, "out": out, "w2c": w2c, "source": src_all,
                "frames": getattr(src_all, "raw", None) if fused_raw else getattr(src_all, "frames", None),
                "ins": self.instance_maps[dataset], "dmap": dmap, "rig": rig, "segments": segments, "fpl": fpl,
                "crop": tuple(float(v) for v in np.asarray(crop).reshape(-1)), "pipe": eng._pipeline(), "eng": eng,
                "fused_raw": fused_raw}
        self._launch_memo[dataset] = memo
        return memo

    def _replay_launches(self, eng, dataset, memo):
        """Issue a memoised launch list.  -> None when all of it went out; on out-of-memory (the library could not grow its
        stamp scratch: CAMA_ENOMEM -> torch.OutOfMemoryError) the memo is dropped, the engine's frames-per-call budget is
        shrunk and the number of FRAMES already issued is returned -- render_clip's halving loop renders the rest."""
        import torch
        launch, desc, issued = eng.render_clip_launch, memo["desc"], 0
        for a in memo["launches"]:
            try:
                launch(desc, *a)
            except torch.OutOfMemoryError:
                self._launch_memo.pop(dataset, None)
                eng.shrink_frames_per_call()
                return issued
            issued += a[1]
        return None

    def _launches_valid(self, memo, eng, dataset, out, w2c, fpl, segments):
        """Everything a memoised launch list was derived from is still the very object it was derived from."""
        src = self._frame_source
        return (memo["out"] is out and memo["w2c"] is w2c and memo["source"] is src and memo["eng"] is eng
                and memo["pipe"] is eng._pipe and memo["fpl"] == fpl and memo["segments"] == segments
                and memo["ins"] is self.instance_maps.get(dataset) and memo["rig"] is self._rig_cache
                and (getattr(src, "raw", None) if memo["fused_raw"] else getattr(src, "frames", None)) is memo["frames"]
                and memo["dmap"] is self._static(dataset)._dmap
                and memo["crop"] == tuple(float(v) for v in self.mm.crop_box()))

    def _pooled_mosaic(self, eng, rig, src_all, ids, step, probe):
        """The mosaic of a whole-clip render the caller did not bring: views of the engine's pooled, placed buffers
        (Engine.pool) -- one tensor, or, for clips beyond MOSAIC_CHUNK_BYTES (8 GiB), a ChunkedMosaic of one buffer per
        launch (each launch's destination is then placed on its own).  `probe`: the source is HBM-resident at output size,
        so candidates can be timed against it."""
        F = len(ids)
        per = rig.C * rig.H * rig.W * 3
        chunk_bytes = int(MOSAIC_CHUNK_BYTES)
        contiguous = ids == list(range(ids[0], ids[0] + F))
        if F * per <= chunk_bytes or step >= F:
            src = src_all.batch(ids) if (probe and contiguous) else None
            return eng.pool.take(eng.mosaic_shape(rig, F), rig, src)
        cuts = list(range(0, F, step)) + [F]
        srcs = [src_all.batch(ids[a:b]) if (probe and contiguous) else None for a, b in zip(cuts, cuts[1:])]
        return ChunkedMosaic(eng.pool.take_many([eng.mosaic_shape(rig, b - a) for a, b in zip(cuts, cuts[1:])], rig, srcs))

    def render_clip(self, dataset, out=None, frames_per_launch=None, poses=None, pipelined=False, segments=None):
        """Render every frame of `dataset` in launches of up to `frames_per_launch` frames.

        Returns (image indices (F,), mosaic device tensor [F, 2H, 3W, 3] uint8).  Nothing is copied to the host.
        out=None: the mosaic is a view of one of the engine's pooled, placed buffers (Engine.pool: recycled once the caller
        lets go of it; a ChunkedMosaic for clips beyond 8 GiB), and HBM-resident frame sources are placed against it once.
        `poses` = a previous frame_poses() result to reuse.  pipelined=True issues the binning and overlay halves on
        two side streams (Engine.render_frames_pipelined) so consecutive launches / clips overlap; the caller must
        then call runtime.engine().join() before consuming `out` on the current stream."""
        import torch
        eng = runtime.engine()
        # segments: the opt-in extension (discs + one-pixel segments between polyline neighbours); None = configs["segments"]
        # ("wu": the anti-aliased variant -- Wu lines blended once by coverage; batched path only)
        segments = _segments_mode(self.configs.get("segments", False) if segments is None else segments)
        idx, w2c = poses if poses is not None else self.frame_poses(dataset)
        resume = 0
        if pipelined and out is not None:
            # the same clip into the same buffers again (a service re-rendering, bench.py's steps): the launches were worked
            # out the first time -- one library call each (Engine.render_clip_launch), nothing else
            memo = self._launch_memo.get(dataset)
            if memo is not None and self._launches_valid(memo, eng, dataset, out, w2c, frames_per_launch, segments):
                done = self._replay_launches(eng, dataset, memo)
                if done is None:
                    return idx, out
                resume = done                                   # out of memory part-way: the loop below takes over from there
        rig = self._rig()
        dmap = self._static(dataset).device()
        F = len(idx)
        shape = eng.mosaic_shape(rig, F)
        if F == 0:
            return idx, (out if out is not None else torch.empty(shape, dtype=torch.uint8, device=eng.device))
        crop = self.mm.crop_box()
        src_all = self.frame_source()
        fused_raw = getattr(src_all, "fused", False) and hasattr(src_all, "raw_batch")
        if fused_raw and segments:
            # (render_vectors falls back to the image-by-image path for this combination; a whole-clip render has no such path)
            raise ValueError("render_clip: the segment extension needs pre-resized frames -- raw sensor frames are resampled inside "
                             "the overlay kernel, which draws discs only (use a non-fused source or segments=False)")
        resident = hasattr(src_all, "frames") or hasattr(src_all, "raw")
        if frames_per_launch:
            step = frames_per_launch
        else:
            # sources that decode / allocate a batch per call (files on disk) count it against the per-call budget
            c0 = self.cm_list[0]
            raw_bytes = rig.C * int(c0.height_origin) * int(c0.width_origin) * 3 if fused_raw else None
            step = eng.max_frames_per_call(dmap, rig, resident_frames=resident, src_bytes_per_frame=raw_bytes,
                                           pipelined=pipelined)
        ids = idx.tolist()
        if out is None:
            out = self._pooled_mosaic(eng, rig, src_all, ids, step, resident and not fused_raw)
        assert tuple(out.shape) == shape
        if resident and not fused_raw and hasattr(src_all, "place_for") and not isinstance(out, ChunkedMosaic):
            src_all.place_for(eng, rig, out, ids)
        # pipelined launches take the host float32 poses as they are (staged by the library); the others one upload
        host_poses = pipelined and isinstance(w2c, np.ndarray) and w2c.dtype == np.float32
        T = w2c if host_poses else eng._mats(w2c)
        # a ChunkedMosaic (one pooled allocation per launch, _pooled_mosaic): launches end at its chunk boundaries
        cuts = sorted(set(getattr(out, "bounds", ())))
        if host_poses and resident and ids == list(range(ids[0], ids[0] + F)) and eng.alpha256 == 256 \
                and LAUNCH_MEMO:
            memo = self._plan_launches(eng, dataset, dmap, rig, src_all, fused_raw, out, w2c, ids, step, cuts, crop, segments,
                                       frames_per_launch)
            if memo is not None and not resume:
                done = self._replay_launches(eng, dataset, memo)
                if done is None:
                    return idx, out
                resume = done
                step = max(1, step // 2) if not frames_per_launch else step
        lo = resume
        while lo < F:
            hi = min([F, lo + step] + [c for c in cuts if c > lo][:1])
            try:
                if fused_raw:      # raw sensor frames: undistort + resize inside the overlay kernel
                    eng.render_frames_raw(dmap, rig, T[lo:hi], src_all.raw_batch(ids[lo:hi]),
                                          self.cm_list, out=out[lo:hi], crop=crop, pipelined=pipelined)
                else:
                    src = src_all.batch(ids[lo:hi])
                    if pipelined:
                        eng.render_frames_pipelined(dmap, rig, T[lo:hi], src, out[lo:hi], crop=crop, segments=segments)
                    else:
                        eng.render_frames(dmap, rig, T[lo:hi], src, out=out[lo:hi], crop=crop, segments=segments)
            except torch.OutOfMemoryError:
                # the per-call scratch is a worst case (every vertex visible in every camera) sized from the memory that
                # was free when the shape was first seen; if the device has filled up since, render fewer frames per call
                if frames_per_launch or step <= 1:
                    raise
                eng.shrink_frames_per_call()
                step = max(1, step // 2)
                continue
            lo = hi
        return idx, out


def clip_mosaics(clips, dataset, poses=None):
    """One mosaic [F, 2H, 3W, 3] per clip for render_clips(), as views of the engine's pooled, placed buffers (Engine.pool.
    take_many: the fastest of one pool of candidate allocations when the clips' frames are HBM-resident; recycled once the
    caller lets go of them).  What a caller of render_clips uses when it does not bring its own buffers."""
    eng = runtime.engine()
    clips = list(clips)
    shapes, srcs, rig0 = [], [], None
    for k, cm in enumerate(clips):
        idx, _ = poses[k] if poses is not None else cm.frame_poses(dataset)
        rig = cm._rig()
        rig0 = rig0 or rig
        shapes.append(eng.mosaic_shape(rig, len(idx)))
        src_all = cm.frame_source()
        ids = idx.tolist()
        ok = (hasattr(src_all, "frames") and not getattr(src_all, "fused", False) and len(ids)
              and ids == list(range(ids[0], ids[0] + len(ids))) and (rig.C, rig.H, rig.W) == (rig0.C, rig0.H, rig0.W))
        srcs.append(src_all.batch(ids) if ok else None)
    return eng.pool.take_many(shapes, rig0, srcs) if clips else []


def render_clips(clips, dataset, outs, pipelined=True, poses=None, max_frames_per_launch=None):
    """Render several clips (scenes) of one dataset pass back to back -- main.py:32's scene loop with the per-frame body
    fused -- into `outs` (one [F, 2H, 3W, 3] device tensor per clip).

    Scenes that agree in frame count, rig geometry and vertex dtype, have pre-resized device-resident frames and maps
    small enough to run without the block index go out as ONE multi-scene launch per group of scenes (Engine.render_scenes:
    one binning chain + one overlay launch for up to `max_frames_per_launch` frames, default 16 384); everything else
    falls back to one ClipManager.render_clip per scene.  Same bytes either way.  With pipelined=True the outputs are
    complete after runtime.engine().join().  `poses`: optional list of frame_poses() results, one per clip."""
    eng = runtime.engine()
    clips = list(clips)
    assert len(outs) == len(clips)
    items = []
    for k, cm in enumerate(clips):
        idx, w2c = poses[k] if poses is not None else cm.frame_poses(dataset)
        src_all = cm.frame_source()
        frames = None
        if hasattr(src_all, "frames") and not getattr(src_all, "fused", False) and len(idx):
            frames = src_all.batch(idx.tolist())
        items.append((cm, idx, w2c, frames))
    batch = [(cm._static(dataset).device(), cm._rig(), w2c, fr, outs[k]) for k, (cm, idx, w2c, fr) in enumerate(items)
             if fr is not None]
    crop0 = np.asarray(clips[0].mm.crop_box(), np.float64).reshape(-1) if clips else None
    same_crop = all(np.array_equal(np.asarray(cm.mm.crop_box(), np.float64).reshape(-1), crop0) for cm in clips)
    if len(batch) == len(clips) and same_crop and eng.scene_batchable(batch):
        F = len(items[0][1])
        from .engine import MAX_SCENES_PER_LAUNCH
        per = max(1, min(MAX_SCENES_PER_LAUNCH, (max_frames_per_launch or 16384) // max(1, F)))
        for lo in range(0, len(batch), per):
            group = batch[lo:lo + per]
            if len(group) == 1:
                cm, idx, w2c, _ = items[lo]
                cm.render_clip(dataset, out=outs[lo], pipelined=pipelined, poses=(idx, w2c))
            else:
                eng.render_scenes(group, crop=clips[0].mm.crop_box(), pipelined=pipelined)
        return True
    for k, (cm, idx, w2c, _) in enumerate(items):
        cm.render_clip(dataset, out=outs[k], pipelined=pipelined, poses=(idx, w2c))
    return False
