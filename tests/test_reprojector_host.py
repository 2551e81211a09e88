"""The config.yaml contract and the Reprojector facade's host side (no GPU): BASELINE.json's north_star names both
("Keep the Reprojector/PoseTransformer class surface and config.yaml contract"); reference: config.yaml:1-25, main.py:21-50."""
import os

import pytest
import yaml

from cama_amd import reprojector
from cama_amd.synth import DEFAULT_CAMA_CONFIGS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLE = os.path.join(REPO, "examples", "config.yaml")


def test_example_config_carries_the_references_keys():
    """examples/config.yaml loads with yaml.safe_load (main.py:24-25) and has every key main.py and the cama classes read
    (config.yaml:1-25): the top-level ones and the cama_configs block, whose values equal the reference's "leave unchanged"
    defaults (= synth.DEFAULT_CAMA_CONFIGS, which the fixtures were generated with)."""
    with open(EXAMPLE) as f:
        raw = yaml.safe_load(f)
    assert set(raw) == set(reprojector.TOP_LEVEL_KEYS)
    assert raw["map_classes"] == ["lane_marking", "Road_teeth", "Crosswalk_Line"]          # config.yaml:14
    assert isinstance(raw["scene_names"], list) and raw["scene_names"]
    cfg = reprojector.load_configs(EXAMPLE)
    assert cfg == raw
    cc = cfg["cama_configs"]
    assert {k: cc[k] for k in reprojector.CAMA_CONFIG_KEYS} == dict(DEFAULT_CAMA_CONFIGS)
    assert set(cc) - set(reprojector.CAMA_CONFIG_KEYS) <= set(reprojector.EXTENSION_KEYS)


def test_load_configs_names_what_is_missing(tmp_path):
    p = tmp_path / "c.yaml"
    p.write_text("version: v1.0-test\n")
    with pytest.raises(KeyError, match="cama_configs"):
        reprojector.load_configs(str(p))
    cc = dict(DEFAULT_CAMA_CONFIGS)
    del cc["pose_prefix"]
    p.write_text(yaml.safe_dump({"cama_configs": cc}))
    with pytest.raises(KeyError, match="pose_prefix"):
        reprojector.load_configs(str(p))
    cc = dict(DEFAULT_CAMA_CONFIGS, camera_main="camera_top")
    with pytest.raises(ValueError, match="camera_main"):
        reprojector.check_cama_configs(cc)
    with pytest.raises(TypeError):
        reprojector.check_cama_configs(["maps"])


def test_reprojector_is_exported_where_the_reference_keeps_the_path():
    """`from cama.reproject import Reprojector` (and the reference's own names next to it)."""
    import cama.reproject as R
    assert R.Reprojector is reprojector.Reprojector and R.load_configs is reprojector.load_configs
    for name in ("BaseManager", "MapManager", "CameraManager"):
        assert hasattr(R, name)


def test_reprojector_accepts_the_whole_config_or_the_cama_block(monkeypatch):
    made = []

    class FakeClip:
        def __init__(self, configs, clip_path=None, output_size=None):
            made.append((configs, clip_path, output_size))
            self.instance_maps = {"nuscenes": []}
            self.output_size = output_size or (540, 960)

    monkeypatch.setattr(reprojector, "ClipManager", FakeClip)
    whole = reprojector.load_configs(EXAMPLE)
    a = reprojector.Reprojector(whole, "/clips/s")
    b = reprojector.Reprojector(whole["cama_configs"], "/clips/s", output_size=(90, 160))
    c = reprojector.Reprojector(EXAMPLE, "/clips/s")
    assert made[0][0] is whole["cama_configs"] and made[1][0] is whole["cama_configs"] and made[1][2] == (90, 160)
    assert made[2][0] == whole["cama_configs"]
    assert a.datasets() == ["nuscenes"] and b.clip_path == "/clips/s" and c.cama_configs == whole["cama_configs"]
