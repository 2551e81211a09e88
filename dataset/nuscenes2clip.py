"""Import-compatible stand-in for the reference's nuScenes -> clip converter (dataset/nuscenes2clip.py).

The converter is OUT OF SCOPE for this build (it needs the nuScenes database,
nuscenes-devkit and shapely, none of which exist on either box, and it is one-off file I/O).  main.py
imports `nuScenes2Clip` unconditionally and calls `convert(scene_name)` before every clip, so this shim
keeps that flow alive for clips that were converted beforehand (by the reference's own converter or by
cama_amd.synth.make_clip): `convert` verifies that the clip directory exists and otherwise explains
what is missing.
"""
import os


class nuScenes2Clip:
    def __init__(self, configs):
        self.configs = configs
        self.output_root = configs["converted_dataroot"]

    def convert(self, scene_name):
        clip = os.path.join(self.output_root, scene_name)
        if os.path.exists(os.path.join(clip, "attribute.json")):
            return clip
        raise RuntimeError(
            f"{clip} is not a converted clip.  The nuScenes -> clip conversion is outside this build's scope; "
            "run the reference's dataset/nuscenes2clip.py once (needs nuscenes-devkit + shapely), or create a "
            "synthetic clip with cama_amd.synth.make_clip().")
