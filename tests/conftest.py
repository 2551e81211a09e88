import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C half of the oracle is test infrastructure; build it on demand (gcc is on both boxes)
    so = os.path.join(REPO, "oracle", "_build", "liboracle.so")
    src = os.path.join(REPO, "oracle", "cama_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def repo_root():
    return REPO
