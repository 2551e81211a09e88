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


MUTATORS = {"": None, "mutate_pose_gaps": mutate_pose_gaps}


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
