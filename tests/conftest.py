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
    config.addinivalue_line("markers", "ab_library: the test's gpu_ctx is a context of the A/B library (-DSKL_AB), the only build that "
                                       "reads the \"A/B only, results identical\" switches (SKL_KNN_SYMMETRIC, SKL_TOPK_STREAM, ...)")


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


@pytest.fixture(autouse=True)
def _switches_follow_the_environment(request):
    """The library reads its SKL_* switches when a context is created; the session-wide context is
    brought back in line with the (restored) environment after every test that changed it."""
    yield
    if "gpu_ctx" in request.fixturenames and "monkeypatch" in request.fixturenames and not request.node.get_closest_marker("ab_library"):
        request.getfixturevalue("_product_ctx").reload_env()


@pytest.fixture()
def set_switch(gpu_ctx, monkeypatch):
    """set_switch("SKL_X", "value" | None): change a library switch for the rest of this test."""
    def _set(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, str(value))
        gpu_ctx.reload_env()
    return _set


@pytest.fixture(scope="session")
def _product_ctx(skl):
    if skl.device_count() == 0:
        pytest.fail("no gfx950 device visible: -m gpu tests must run on the GPU box")
    ctx = skl.Context(0)
    # The library's default tie rule is the reference binary's (SKL_KNN_TIES_REFERENCE, tests/test_gpu_knn_ties.py); the
    # session-wide context runs canonical ties -- a property of the data alone -- unless a test sets the mode itself.
    ctx.set_knn_ties(skl.TIES_CANONICAL)
    yield ctx
    ctx.close()


@pytest.fixture()
def gpu_ctx(request, skl, _product_ctx):
    """The session-wide context of the PRODUCT library -- or, for a test marked `ab_library`, a context of the A/B library
    for the length of that test, during which every call of the binding goes to that build (it re-reads its switches at
    every entry point, so monkeypatched SKL_* variables take effect at once)."""
    if request.node.get_closest_marker("ab_library") is None:
        yield _product_ctx
        return
    import sketchlib.rust_amd as pkg

    with skl.using_library(pkg.build_ab_library()):
        ctx = skl.Context(0)
        ctx.set_knn_ties(skl.TIES_CANONICAL)
        try:
            yield ctx
        finally:
            ctx.close()
