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


@pytest.fixture(params=["default", "generic"])
def rs_kernel(request, monkeypatch):
    """the tile-chain kernel behind `resample owed -> effects.lowpass / highpass`: k_rsp (rs_periodic.hip) where the rate and the rows allow it — int16 rows,
    cubic, 44.1 / 22.05 kHz -> 48 kHz — else k_rs_onepole (flac_tail.hip); "generic" switches k_rsp off so that k_rs_onepole keeps its tests at those rates"""
    if request.param == "generic":
        monkeypatch.setenv("AUKIT_RS_GENERIC", "1")
    return request.param
