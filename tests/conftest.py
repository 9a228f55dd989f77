import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """libpnr.so travels prebuilt to the GPU box; build it (and the C oracle) if absent."""
    so = os.path.join(ROOT, "panopticnerf_amd", "libpnr.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "panopticnerf_amd", "csrc"), "-j8"])
    from oracle import c_oracle
    c_oracle.build()


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "path_small.npz")))


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    return torch.device("cuda:0")
