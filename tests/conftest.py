import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: extra-coverage cases kept out of the driver's time-limited run; ISST_RUN_SLOW=1 runs them")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a device: on a box without one they are skipped (the driver selects them with -m gpu on the GPU box)."""
    import torch
    if os.environ.get("ISST_RUN_SLOW", "0") in ("", "0"):
        slow = pytest.mark.skip(reason="slow extra-coverage case (ISST_RUN_SLOW=1 runs it)")
        for item in items:
            if "slow" in item.keywords:
                item.add_marker(slow)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible (the hot path has no CPU implementation)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_library():
    """The HIP library is built in-tree (git-ignored).  A fresh checkout has none: build it once (hipcc cross-compiles for gfx950
    without a GPU, ~2 min) so that the ABI tests check a real library instead of failing on a missing file."""
    so = os.path.join(ROOT, "infinisst_amd", "libinfinisst_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    return so


@pytest.fixture(autouse=True)
def _library_for_tests_that_load_it(request):
    """Only the GPU tests and the ABI / launcher tests dlopen the library; the oracle / golden / host-logic tests run without hipcc."""
    if "gpu" in request.keywords or request.module.__name__ in ("test_abi_and_host", "test_streams_gloo"):
        request.getfixturevalue("built_library")
