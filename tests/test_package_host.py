"""cama_amd/__init__.py: the one process-wide default the package sets for the HIP runtime (GPU_MAX_HW_QUEUES)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    e.update(env)
    e["PYTHONPATH"] = REPO + os.pathsep + e.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)


def test_import_sets_eight_hardware_queues_unless_the_caller_chose():
    code = "import os, cama; import cama_amd; print(os.environ['GPU_MAX_HW_QUEUES']); print(cama_amd.hw_queue_default())"
    r = _run(code)
    assert r.returncode == 0, r.stderr
    value, note = r.stdout.strip().splitlines()
    assert value == "8" and note == "GPU_MAX_HW_QUEUES=8 set by cama_amd"
    r = _run(code, GPU_MAX_HW_QUEUES="4")                      # the caller's choice stands
    assert r.returncode == 0, r.stderr
    value, note = r.stdout.strip().splitlines()
    assert value == "4" and note == "caller's GPU_MAX_HW_QUEUES=4 kept"


def test_the_reference_import_path_sets_it_before_torch_is_touched():
    """main.py imports `cama.tools` and `cama.dataset` at its top (main.py:6-7): the variable must be in the
    environment by the time that import returns, whether or not torch was imported before (it is read at the first HIP call)."""
    r = _run("import torch, os; assert 'GPU_MAX_HW_QUEUES' not in os.environ; from cama.dataset import ClipManager; "
             "print(os.environ['GPU_MAX_HW_QUEUES'])")
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "8"
