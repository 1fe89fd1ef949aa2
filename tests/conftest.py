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
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def ctx():
    """A GPU context; the HIP library must be present and must be the thing that runs (no fallback)."""
    from aukit_amd import _native as N
    from aukit_amd import batch as B
    if not os.path.exists(N.LIB_PATH):
        pytest.fail("aukit_amd/libaukit_hip.so is missing: run __graft_entry__.build() first")
    c = B.Context(0)
    yield c
    c.close()
