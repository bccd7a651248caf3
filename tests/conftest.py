import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """The HIP library is built in-tree (git-ignored).  A fresh checkout has none: build it once (hipcc cross-compiles for gfx950
    without a GPU, ~2 min) so that the ABI tests check a real library instead of failing on a missing file."""
    so = os.path.join(ROOT, "infinisst_amd", "libinfinisst_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return so
