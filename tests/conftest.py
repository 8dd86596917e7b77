import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libssvio_ref.so (the compiled reference)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle bindings (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def ref_available(po):
    return po.have_ref()


@pytest.fixture(scope="session")
def ctx():
    """One ssx context on GPU 0.  Fails loudly (no skip, no fallback) when the HIP path is unavailable."""
    import ssvio_amd
    c = ssvio_amd.Context(0)
    yield c
    c.close()
