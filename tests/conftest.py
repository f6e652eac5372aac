import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pumipic_amd_loader  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ppo():
    """The CPU oracle (test infrastructure)."""
    return pumipic_amd_loader.load_oracle()


@pytest.fixture(scope="session")
def pp():
    """The product package."""
    return pumipic_amd_loader.load()


@pytest.fixture(scope="session")
def synth(pp):
    return pp.synth


@pytest.fixture(autouse=True)
def _no_hip_error_left_behind(request):
    """after every GPU test: the HIP runtime's sticky last error must be clear -- a launch that failed
    unnoticed would otherwise surface in whatever checks next (RCCL checks after its own launches)"""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    try:
        from pumipic_amd import capi
        lib = capi.lib()
    except Exception:  # noqa: BLE001 -- library not built / not loadable: other tests say so
        return
    code, msg = capi.peek_hip_error()
    assert code == 0, "HIP error left behind by %s: %s" % (request.node.nodeid, msg)
