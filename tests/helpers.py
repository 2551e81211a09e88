"""Shared test helpers: rebuild the synthetic clip a golden fixture was captured on."""
import json
import os
from os.path import join

import numpy as np

from cama_amd.synth import make_clip, DEFAULT_CAMA_CONFIGS, CAMERA_NAMES

GOLDEN = join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLIP_TAGS = ["a", "b_exact", "c_gaps", "d_nusonly", "e_crop"]


def mutate_pose_gaps(clip):
    # must mirror tests/golden/gen_golden.py:mutate_pose_gaps
    for name in ("scmv_camera_front.txt", "wigo_offset_clip.txt"):
        p = join(clip, "odometry", name)
        rows = np.loadtxt(p)
        keep = np.ones(len(rows), bool)
        keep[3] = False
        keep[-3:] = False
        np.savetxt(p, rows[keep])


def mutate_single_point_labels(clip):
    # must mirror tests/golden/gen_golden.py:mutate_single_point_labels
    rng = np.random.default_rng(77)
    c, s = np.cos(0.3), np.sin(0.3)
    p = join(clip, "maps", "map_labels.json")
    if os.path.exists(p):
        labels = json.load(open(p))
        for k in range(40):
            a, l = rng.uniform(2.0, 45.0), rng.uniform(-7.0, 7.0)
            x, y = -290.0 + a * c - l * s, -280.0 + a * s + l * c
            px, py = (y + 300.0) / 0.1, (x + 300.0) / 0.1
            ang = rng.uniform(0, 2 * np.pi)
            L = rng.uniform(0.105, 0.195)
            labels.append({"attrs": {"type": ["lane_marking", "Road_teeth", "Crosswalk_Line"][k % 3]},
                           "data": [[px, py], [px + L * np.cos(ang), py + L * np.sin(ang)]], "id": 9100 + k})
        json.dump(labels, open(p, "w"))
    p = join(clip, "maps", "map_nuscenes.json")
    if os.path.exists(p):
        labels = json.load(open(p))
        for k in range(40):
            x, y = rng.uniform(-25.0, 45.0), rng.uniform(-9.0, 9.0)
            ang = rng.uniform(0, 2 * np.pi)
            L = rng.uniform(0.105, 0.195)
            labels.append({"attrs": {"type": ["Road_teeth", "lane_marking", "Stop_Line"][k % 3]},
                           "data": [[x, y], [x + L * np.cos(ang), y + L * np.sin(ang)]], "id": 9200 + k})
        json.dump(labels, open(p, "w"))


MUTATORS = {"": None, "mutate_pose_gaps": mutate_pose_gaps, "mutate_single_point_labels": mutate_single_point_labels}


def single_point_deviation(g, ds, frame_ids, project):
    """Compare a flat projector with the reference's golden (v,u) on a fixture that contains one-point instances.
    `project(idx)` -> per camera: visible (v,u) float64 in draw order.  Returns (max |dev| over points of multi-point
    instances, max |dev| over points of ONE-point instances, number of one-point projections compared,
    number of truncated-pixel flips)."""
    dev_multi = dev_single = 0.0
    n_single = flips = 0
    for idx in frame_ids:
        got_all = project(idx)
        for ci, name in enumerate(CAMERA_NAMES):
            gl = golden_instances(g, f"{ds}_f{idx}_{name}_vu")
            gold = np.concatenate([p for _, p in gl]).reshape(-1, 2) if gl else np.zeros((0, 2))
            got = got_all[ci]
            assert got.shape == gold.shape, (ds, idx, name, got.shape, gold.shape)
            single = np.repeat(np.asarray([len(p) == 1 for _, p in gl], bool), [len(p) for _, p in gl]) \
                if gl else np.zeros(0, bool)
            d = np.abs(got - gold).max(axis=1) if len(gold) else np.zeros(0)
            if (~single).any():
                dev_multi = max(dev_multi, float(d[~single].max()))
            if single.any():
                dev_single = max(dev_single, float(d[single].max()))
                n_single += int(single.sum())
            flips += int((got.astype(np.int32) != gold.astype(np.int32)).any(axis=1).sum())
    return dev_multi, dev_single, n_single, flips


def load_golden(tag):
    return np.load(join(GOLDEN, f"clip_{tag}.npz"))


def rebuild_clip(g, tmpdir, **overrides):
    kw = json.loads(str(g["clip_kwargs"]))
    kw.update(overrides)
    clip = join(str(tmpdir), "clip")
    make_clip(clip, **kw)
    mut = MUTATORS[str(g["mutate"])]
    if mut is not None:
        mut(clip)
    return clip


def golden_instances(g, prefix):
    """inverse of gen_golden.pack_instances -> list of (class, points)."""
    classes = [str(c) for c in g[prefix + "_classes"]]
    counts = g[prefix + "_counts"]
    pts = g[prefix + "_points"]
    out, o = [], 0
    for c, n in zip(classes, counts):
        out.append((c, pts[o:o + n]))
        o += n
    return out


def assert_instances_equal(instances, golden_list, exact=True, atol=0.0):
    assert len(instances) == len(golden_list), (len(instances), len(golden_list))
    for ins, (cls, pts) in zip(instances, golden_list):
        assert ins["class"] == cls
        p = np.asarray(ins["points"])
        assert p.shape == pts.shape, (p.shape, pts.shape)
        if exact:
            assert np.array_equal(p, pts), float(np.abs(p - pts).max())
        else:
            assert np.allclose(p, pts, rtol=0, atol=atol), float(np.abs(p - pts).max())
