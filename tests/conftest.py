import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "deep-tracking-control_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The tests exercise libdtc_hip.so through its C ABI; (re)build it in-tree when it is missing or older than its
    sources (hipcc cross-compiles gfx950 without a GPU).  The product itself never builds or falls back: a missing
    library raises DtcError (tests/test_abi_and_host.py)."""
    import build as dtc_build
    dtc_build.build(verbose=False)
