"""-m gpu: bench.py as the driver runs it -- a subprocess, the JSON line on stdout.

  * `python bench.py --gpus 2` (no torchrun, no WORLD_SIZE) must start its own ranks.  The box has one
    GPU, so the two ranks share it (CAMA_BENCH_SHARE_GPU=1 -> gloo for the one collective: RCCL refuses two ranks on one
    device); what is checked is the N > 1 code path: sharding, the all_gather, the per-scene hash check.
  * the N = 1 line: no roofline fraction above 1 on a map where the vertex term matters, the
    projection has its own roofline object, a sustained figure and the CPU baselines are present.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=1500):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, cwd=REPO, env=env, capture_output=True,
                       text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_launches_its_own_ranks():
    env = {"CAMA_BENCH_SHARE_GPU": "1"}
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    p, line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], env)
    assert p.returncode == 0, p.stderr[-3000:]
    assert line is not None, p.stdout[-2000:]
    assert line["n_gpus"] == 2 and line["rccl_world"] == 2
    assert line["scaling"] == "strong" and line["config"]["scenes"] == 73
    chk = line["hash_check"]
    assert chk["verified"] == 73 and not chk["mismatched"] and not chk["missing"] and not chk["unverified"]
    assert len(line["per_rank_frames"]) == 2 and sum(line["per_rank_frames"]) == 73 * 40 * 2
    st = line["stress"]
    assert st["hash_check"]["verified"] == 16 and not st["hash_check"]["mismatched"]
    for r in (line["roofline"], line["roofline_project"], st["roofline"], st["roofline_project"]):
        assert 0.0 < r["frac"] <= 1.0, r
    # every rank pinned itself to its own cores (disjoint contiguous shares of the allowed / NUMA-local ones)
    aff = line["rank_affinity"]
    assert aff["bound"] == [True, True] and min(aff["cpus_per_rank"]) >= 1
    spans = sorted(zip(aff["first_cpu"], aff["last_cpu"]))
    assert spans[0][1] < spans[1][0], aff
    # ranks sharing one GPU run without placement auditions; the line says where placement lives
    assert line["placement"]["source"].startswith("off") and line["placement"]["pool"]["auditions"] == 0
    # ... and that a rank's transient placement memory is bounded: none at all when ranks share a GPU
    assert line["placement"]["audition_peak_bytes"] == 0 and line["placement"]["audition_ms"] == 0.0
    assert "reference_default" not in line and "ingest" not in line            # nested N = 1 measurements only


def test_bench_gpus_beyond_the_node_is_refused_not_asserted():
    """Without the share switch, asking for more GPUs than the node has is a clear non-zero exit, not an AssertionError
    from deep inside a rank."""
    import torch
    n = torch.cuda.device_count()
    p, line = _run(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0"], {"CAMA_BENCH_SHARE_GPU": "0"}, timeout=300)
    assert p.returncode == 2 and line is None
    assert "GPU(s) visible" in p.stderr


def test_bench_line_site_map_roofline_is_attributed_per_kernel():
    """A site-sized map (~3e5 vertices over the 600 m extent): the overlay is charged its image bytes only, the
    projection its own (culled) vertex bytes -- no fraction above 1 anywhere, whole-step below the kernel figures."""
    p, line = _run(["--map", "site", "--verts", "300000", "--frames", "16", "--steps", "6", "--warmup", "2",
                    "--cpu-seconds", "0", "--no-verify", "--sustain-seconds", "0.3"])
    assert p.returncode == 0, p.stderr[-3000:]
    ro, rp = line["roofline"], line["roofline_project"]
    assert ro["kernel"] == "k_overlay" and ro["bytes_per_launch"] == 36 * 1600 * 900 * 16
    assert 0.3 < ro["frac"] <= 1.0
    assert rp["launches"] > 0 and 0.0 < rp["frac"] <= 1.0
    assert 0.0 < rp["vertex_read_fraction"] < 0.5              # most of the site is outside the crop box: never fetched
    assert line["hbm_frac_whole_step"] <= max(ro["frac"], rp["frac"]) + 1e-9
    assert line["projection_stats"]["block_cull"] is True
    assert line["sustained"]["seconds"] >= 0.27 and line["sustained"]["value"] > 0        # (about 0.3 s: sized from a calibration region)


def test_bench_default_line_has_every_contract_field():
    p, line = _run(["--steps", "20", "--warmup", "5", "--cpu-seconds", "3", "--cpu-pool-seconds", "2", "--cpu-workers", "4"])
    assert p.returncode == 0, p.stderr[-3000:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert line["hash_check"]["verified"] == 1
    assert line["roofline"]["bytes_per_launch"] == 36 * 1600 * 900 * 40
    assert 0.5 < line["roofline"]["frac"] <= 1.0
    assert line["sustained"]["seconds"] >= 0.9
    cb = line["cpu_baseline"]
    assert cb["cores"] == 1 and cb["value"] > 0
    assert cb["all_cores"]["cores"] == 4 and cb["all_cores"]["value"] > 0
    # the headline's launches touch 2 GB: by the end of the run the library has timed both workgroup -> band orders on
    # this process's own launches and settled on one of them
    om = line["overlay_mapping"]
    assert om["decided"] in (5, 31) and min(om["samples"]) >= 2 and min(om["ns_per_mb"]) > 0, om
    # `value` is the exactly-K-steps region (the sustained region is printed beside it, never instead of it)
    assert abs(line["value"] - 20 * 40 / (line["ms_per_step"] * 20e-3)) < 1e-6 * line["value"]
    assert abs(line["per_rank_seconds"][0] - line["ms_per_step"] * 20e-3) < 1e-9
    # placement: at most 16 mosaic candidates, its cost is in the line, and on a flat box it keeps the plain allocation
    pl = line["placement"]
    assert pl["candidates_per_mosaic"] <= 16 and pl["audition_ms"] > 0
    assert pl["audition_peak_bytes"] <= 16 * 36 * 1600 * 900 * 41
    assert pl["verdict"] in ("fast placement found", "no fast mode on this box", "budget")
    if pl["flat_box"]:
        assert pl["candidates_per_mosaic"] == 8 and pl["candidates_per_frames"] == 0 and pl["kept_of_pool"] == 1
    # the reference's default pipeline (raw 1600x900 -> 960x540) and the ingest stage, nested in the driver-run line
    rd = line["reference_default"]
    assert rd["hash_check"]["verified"] == 1 and rd["hash_check"]["timed_output_equals_verified_render"]
    assert rd["roofline"]["kernel"] == "k_overlay_raw35" and rd["roofline"]["bytes_per_launch"] == 18 * (1600 * 900 + 960 * 540) * 40
    assert 0.3 < rd["roofline"]["frac"] <= 1.0 and rd["value"] > 0 and rd["hbm_frac_whole_step"] <= rd["roofline"]["frac"] + 0.02
    ing = line["ingest"]
    assert ing["byte_equal_to_host_decoder"] is True and ing["value"] > 1000 and ing["bytes_out_per_image"] == 1600 * 900 * 3
    assert ing["decoder_stats"]["host_unsupported"] == 0 and ing["decoder_stats"]["host_flagged"] == 0
    fl = ing["images_per_s_by_batches_in_flight"]                      # one, two, three batches at once (the pump's steady state)
    assert set(fl) == {"1", "2", "3"} and min(fl.values()) > 1000
    assert rd["leg_seconds"] + ing["leg_seconds"] < 12.0


def test_bench_timed_path_that_renders_nothing_is_caught():
    """the output buffers are poisoned between the warm-up and the timed region, so the hash check after
    it can only pass on bytes the timed steps wrote.  Fault injection: the timed steps render nothing -> exit 3, no line."""
    args = ["--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--sustain-seconds", "0", "--no-extras"]
    p, line = _run(args, {"CAMA_BENCH_FAULT": "skip_overlay"}, timeout=600)
    assert p.returncode == 3 and line is None, (p.returncode, p.stdout[-500:], p.stderr[-1500:])
    assert "timed path's output differs" in p.stderr
    # the same command without the fault prints its line (the poison itself breaks nothing)
    p, line = _run(args, timeout=600)
    assert p.returncode == 0 and line["hash_check"]["verified"] == 1, p.stderr[-1500:]
    ro = line["roofline"]
    assert 0.0 < ro["launch_ms_min"] <= ro["avg_launch_ms"] <= ro["launch_ms_max"]
    assert line["scratch_bytes"] > 0


def test_bench_frame_sharded_timed_path_is_checked_too():
    """The frame-sharded (stress-shaped) job: same poison + fault, on the sampled frame positions."""
    args = ["--map", "random", "--verts", "200000", "--frames", "48", "--shard-frames", "--height", "180", "--width", "320",
            "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--sustain-seconds", "0"]
    p, line = _run(args, {"CAMA_BENCH_FAULT": "skip_overlay"}, timeout=600)
    assert p.returncode == 3 and line is None, (p.returncode, p.stderr[-1500:])
    p, line = _run(args, timeout=600)
    assert p.returncode == 0 and line is not None, p.stderr[-1500:]


def test_bench_one_rank_rccl_group():
    """the RCCL path of the bench inside the driver-run suite -- a 1-rank `nccl` process group
    (init with device_id, barrier, the int64 all_gather_into_tensor of the report) on the box's one GPU."""
    env = {"CAMA_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517", "RANK": "0", "WORLD_SIZE": "1",
           "LOCAL_RANK": "0", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    p, line = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0", "--sustain-seconds", "0", "--no-extras"], env,
                   timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["rccl_world"] == 1 and "RCCL" in line["collective"], line["collective"]
    assert line["hash_check"]["verified"] == 1
