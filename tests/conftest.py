import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build whatever native piece is missing (seconds; hipcc cross-compiles without a GPU)
    for sub in ("jet-pbrt_amd/csrc", "jet-pbrt_amd/host", "oracle"):
        subprocess.run(["make", "-s"], cwd=os.path.join(REPO, sub), check=True)
    if os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s"], cwd=os.path.join(REPO, "oracle", "ref_build"), check=True)


@pytest.fixture(scope="session")
def H():
    import harness
    return harness


@pytest.fixture(scope="session")
def kat():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "kat.npz"))


@pytest.fixture(scope="session")
def gpu_ctx():
    import jet_pbrt_amd as jp
    ctx = jp.Context(0)
    yield ctx
    ctx.close()
