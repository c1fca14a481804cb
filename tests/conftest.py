import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; never used by the product)."""
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The real reference compiled into oracle/_ref (skips where it was not built)."""
    from oracle.oracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libvisma_ref.so not built (needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def lib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from visma_amd import build
    build.build_lib()
    build.build_experiments()          # the side build (test seam, experiments): built once, before any worker process
    from visma_amd import _lib
    _lib.load()
    return _lib


def _gpu_present():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def _gpu_ctx_session(lib):
    """A HIP context on device 0.  On a GPU box a failure here is a FAILURE, not a skip.
    Pinned to the fp32 search: these tests check the kernels against their fp32 specification
    (oracle group B) bit for bit, and the brute-force kernel against the grid.  The default
    ("auto": f64 search for small clouds) is covered by tests/test_search_precision.py."""
    ctx = lib.Context(0)
    ctx.set_search_precision("f32")
    return ctx


@pytest.fixture(params=["brute", "grid"])
def gpu_ctx(request, lib, _gpu_ctx_session):
    """The session context with the NN search forced to each implementation in
    turn: the brute-force kernel and the radius-cell grid must give identical answers."""
    ctx = _gpu_ctx_session
    ctx.set_nn_mode({"brute": lib.NN_BRUTE, "grid": lib.NN_GRID}[request.param])
    ctx.nn_mode_name = request.param
    yield ctx
    ctx.set_nn_mode(lib.NN_AUTO)


@pytest.fixture(scope="session")
def _gpu_ctx_exact_session(lib):
    """A context with the DEFAULT search: exact (fp32 ranking, f64 re-rank of the rounding band)."""
    return lib.Context(0)


@pytest.fixture(params=["brute", "grid", "exact"])
def gpu_ctx_any(request, lib, _gpu_ctx_session, _gpu_ctx_exact_session):
    """Every search the library has: the two fp32-specification kernels (brute force, grid) and the
    default exact grid search.  `ctx.exact` tells the test which bar applies: the fp32 kernels may
    decide a near-tie or a pair at the radius differently from the reference (a few in 1e5 queries);
    the exact search may not."""
    if request.param == "exact":
        ctx = _gpu_ctx_exact_session
        ctx.set_nn_mode(lib.NN_AUTO)
    else:
        ctx = _gpu_ctx_session
        ctx.set_nn_mode({"brute": lib.NN_BRUTE, "grid": lib.NN_GRID}[request.param])
    ctx.nn_mode_name = request.param
    ctx.exact = request.param == "exact"
    yield ctx
    ctx.set_nn_mode(lib.NN_AUTO)


@pytest.fixture()
def gpu_ctx_auto(lib, _gpu_ctx_session):
    _gpu_ctx_session.set_nn_mode(lib.NN_AUTO)
    return _gpu_ctx_session
