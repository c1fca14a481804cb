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
    from visma_amd import _lib
    _lib.load()
    return _lib


def _gpu_present():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def gpu_ctx(lib):
    """A HIP context on device 0.  On a GPU box a failure here is a FAILURE, not a skip."""
    return lib.Context(0)
