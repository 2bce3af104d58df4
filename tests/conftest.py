import os
import sys
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "sw-nerf_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libswnerf_hip.so is built in-tree (it travels to the GPU box with the snapshot); rebuild it here if it
    is missing or older than its sources, so a fresh checkout can run either suite directly."""
    import __graft_entry__
    __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import cases

    def load(name):
        return dict(np.load(os.path.join(cases.GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    return load
