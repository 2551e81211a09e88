"""`Reprojector`: the one-object view of the reprojection path that BASELINE.json's north_star names ("keep the
Reprojector/PoseTransformer class surface and config.yaml contract").

The reference itself has no class of that name: main.py:50-61 spells the path out as four calls on a
ClipManager and a VideoGenerator.  This facade is exactly those calls under the north_star's name, over the same
ClipManager -- nothing is computed here:

    rp = Reprojector(configs, clip_path)                 # configs: the loaded config.yaml (or its "cama_configs" part)
    for image_idx, mosaic in rp.frames("cama"):          # main.py:57-61 with the body folded in
        video.add_frame(mosaic)
    maps_2d = rp.project(instance_map)                   # ClipManager.project_all_camera   (cama/dataset.py:108-117)
    images  = rp.render(maps_2d, image_idx)              # ClipManager.render_vectors        (cama/dataset.py:119-126)

and `load_configs(path)` is main.py:24-25 (yaml.safe_load) plus a check of the contract's keys (config.yaml:1-25).
"""
import os

from .dataset import ClipManager
from .tools import VideoGenerator

# the reference's config.yaml contract (config.yaml:1-25): top-level keys main.py reads, and the cama_configs block every
# ClipManager / CameraManager / DatasetReader call site indexes
TOP_LEVEL_KEYS = ("version", "dataroot", "converted_dataroot", "scene_names", "cama_label_file", "output_video_dir",
                  "map_classes", "cama_configs")
CAMA_CONFIG_KEYS = ("result_dir", "camera_list", "camera_main", "height_mlp", "pose_prefix", "cama_map_file",
                    "nuscenes_map_file")
# keys this build adds (all optional, all documented in INTEGRATION.md; none changes a rendered byte except "segments",
# which is an extension without reference semantics)
EXTENSION_KEYS = ("output_size", "render_ahead", "device_map_build", "egress", "segments")


def check_cama_configs(cc):
    """The cama_configs block as the reference's classes index it; raises KeyError naming what is missing."""
    if not isinstance(cc, dict):
        raise TypeError("cama_configs must be a mapping (config.yaml:17-25)")
    missing = [k for k in CAMA_CONFIG_KEYS if k not in cc]
    if missing:
        raise KeyError(f"cama_configs lacks {missing} (config.yaml:17-25)")
    if cc["camera_main"] not in cc["camera_list"]:
        raise ValueError(f"camera_main {cc['camera_main']!r} is not in camera_list")
    return cc


def load_configs(path):
    """main.py:24-25: yaml.safe_load of the configuration file, with the contract's keys checked."""
    import yaml
    with open(path, "r") as f:
        configs = yaml.safe_load(f)
    if not isinstance(configs, dict) or "cama_configs" not in configs:
        raise KeyError(f"{path}: no `cama_configs` block (config.yaml:17)")
    check_cama_configs(configs["cama_configs"])
    return configs


class Reprojector:
    def __init__(self, configs, clip_path, output_size=None):
        """configs: the whole loaded config.yaml (its "cama_configs" block is used, main.py:50), that block itself, or a
        path to a yaml file.  clip_path: a converted clip directory (main.py:49)."""
        if isinstance(configs, (str, os.PathLike)):
            configs = load_configs(configs)
        self.configs = configs
        cc = configs["cama_configs"] if "cama_configs" in configs else configs
        self.cama_configs = check_cama_configs(cc)
        self.clip_path = clip_path
        self.clip = ClipManager(self.cama_configs, clip_path, output_size=output_size)

    # ------------------------------------------------------------------ what main.py's loop body calls
    def datasets(self):
        """Which label sets the clip carries: a subset of ("cama", "nuscenes") (cama/dataset.py:13-24)."""
        return [d for d in ("cama", "nuscenes") if d in self.clip.instance_maps]

    def yield_frame(self, dataset):
        return self.clip.yield_frame(dataset=dataset)

    def project(self, instance_map):
        return self.clip.project_all_camera(instance_map)

    def render(self, maps_2d_dict, image_idx):
        return self.clip.render_vectors(maps_2d_dict, image_idx)

    @staticmethod
    def mosaic(image_dict):
        """VideoGenerator.concate_image without an encoder (cama/tools.py:22-25)."""
        return VideoGenerator.concate_image(None, image_dict)

    def frames(self, dataset):
        """main.py:57-60 as one generator: (image_idx, 2x3 mosaic) per renderable frame.  On the fused path the mosaic is
        what VideoGenerator.concate_image returns there: an ndarray view of the batch's pinned host copy."""
        for image_idx, instance_map in self.clip.yield_frame(dataset=dataset):
            yield image_idx, self.mosaic(self.render(self.project(instance_map), image_idx))

    def write_video(self, dataset, output_video_path, sink=None):
        """main.py:55-61: one reprojection video of `dataset`; returns the number of frames written."""
        H, W = self.clip.output_size
        vg = VideoGenerator(output_video_path, output_shape=(3 * W, 2 * H), sink=sink)
        n = 0
        try:
            for image_idx, instance_map in self.clip.yield_frame(dataset=dataset):
                image_dict = self.render(self.project(instance_map), image_idx)
                vg.add_frame(vg.concate_image(image_dict))
                n += 1
        finally:
            vg.close()
        return n

    # ------------------------------------------------------------------ whole-clip device path (extension)
    def render_clip(self, dataset, **kw):
        """ClipManager.render_clip: (image indices, mosaics [F, 2H, 3W, 3] in HBM)."""
        return self.clip.render_clip(dataset, **kw)
