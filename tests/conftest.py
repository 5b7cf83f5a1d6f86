import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FIXTURES = os.path.join(GOLDEN, "reference_fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def skl():
    """ctypes binding of the product library; built in-tree if stale."""
    import sketchlib.rust_amd as pkg
    from sketchlib.rust_amd import capi

    pkg.build_library()
    capi.load()
    return capi


@pytest.fixture(scope="session")
def gpu_ctx(skl):
    if skl.device_count() == 0:
        pytest.fail("no gfx950 device visible: -m gpu tests must run on the GPU box")
    ctx = skl.Context(0)
    yield ctx
    ctx.close()
