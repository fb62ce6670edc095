import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, f"golden_{name}.npz")))


@pytest.fixture(scope="session")
def oracle64():
    from oracle.oracle import Oracle
    return Oracle("f64")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.oracle import Oracle
    return Oracle("f32")


# Guard bands around every device allocation of the library (include/tcsfm.h tcsfm_debug_check_guards): the GPU tests run with them on, and
# every test ends with a check that no kernel wrote outside its buffers (GPU AddressSanitizer is not available on the pool).
# TCSFM_DEBUG_GUARDS=0 in the environment switches them off.
os.environ.setdefault("TCSFM_DEBUG_GUARDS", "1")


@pytest.fixture(autouse=True)
def _guard_bands_intact(request):
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from tightly_coupled_sfm_amd import _lib
    if _lib._lib is None:          # (this test never loaded the library)
        return
    n, bad = _lib.check_guards()
    assert bad == 0, f"{bad} of {n} device allocations have a damaged guard band (details on stderr)"
