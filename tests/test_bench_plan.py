"""`python bench.py --gpus N --plan`: a no-hardware readiness check of the multi-GPU job -- per rank the
scenes, the resident bytes and the stamp scratch, against one MI355X's HBM.  Runs on the CPU box."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plan(argv):
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv + ["--plan"], cwd=REPO, capture_output=True,
                       text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_plan_of_the_8_gpu_job_fits_and_covers_every_scene():
    p, d = _plan(["--gpus", "8"])
    assert p.returncode == 0 and d["plan"] is True and d["fits"] is True, p.stderr[-2000:]
    assert d["gpus"] == 8 and len(d["ranks"]) == 8 and d["scenes"] == 73
    assert sorted(s for r in d["ranks"] for s in r["scenes"]) == list(range(73))
    assert [len(r["scenes"]) for r in d["ranks"]] == [10, 9, 9, 9, 9, 9, 9, 9]
    for r in d["ranks"]:
        assert r["fits"] and r["total_bytes"] < 0.92 * d["hbm_bytes_per_gpu"]
        assert r["resident_frames_bytes"] == len(r["scenes"]) * 41 * 6 * 900 * 1600 * 3
        # the placement auditions' transient candidates are part of the plan and fit beside the frames
        # (round 6: the buffers themselves + at most 16 spare candidates, whatever the number of scenes)
        assert r["placement_transient_bytes"] == 16 * 40 * 6 * 900 * 1600 * 3
        assert r["peak_bytes_during_placement"] <= r["total_bytes"] < 0.92 * d["hbm_bytes_per_gpu"]
        st = r["stress"]                                                   # the nested configs[4] measurement
        assert st["frame_range"][1] - st["frame_range"][0] == 125
        assert st["scratch_held_planned"] < st["scratch_worst_case_one_slot"]     # demand-sized, not 24 B per (f, c, v)
    assert [r["stress"]["frame_range"] for r in d["ranks"]][:2] == [[0, 125], [125, 250]]


def test_plan_refuses_a_job_that_cannot_fit():
    p, d = _plan(["--gpus", "1", "--scenes", "600", "--no-scene-batch"])      # 600 scenes x 1 GB of frames on one GPU
    assert p.returncode == 1 and d["fits"] is False and not d["ranks"][0]["fits"]
    p, d = _plan(["--gpus", "1", "--map", "random", "--verts", "1000000", "--frames", "1000", "--shard-frames"])
    assert p.returncode == 0 and d["ranks"][0]["frame_range"] == [0, 1000] and d["ranks"][0]["frames_per_launch"] == 128


def test_segment_modes_and_workload_keys():
    """configs["segments"] / --segments [--wu]: False, True (Bresenham) or "wu" (anti-aliased); the three render different bytes,
    so their golden-hash keys must differ."""
    import pytest
    import bench
    from cama_amd.dataset import _segments_mode
    assert _segments_mode(False) is False and _segments_mode(None) is False and _segments_mode(0) is False
    assert _segments_mode(True) is True and _segments_mode(1) is True and _segments_mode("bresenham") is True
    assert _segments_mode("wu") == "wu" and _segments_mode("WU") == "wu" and _segments_mode("false") is False
    with pytest.raises(ValueError):
        _segments_mode("thick")
    keys = {bench.args_key(bench.parse_args(a)) for a in ([], ["--segments"], ["--segments", "--wu"])}
    assert len(keys) == 3
    assert bench._segments(bench.parse_args(["--segments", "--wu"])) == "wu" and bench._segments(bench.parse_args(["--segments"])) is True
