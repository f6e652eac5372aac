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
