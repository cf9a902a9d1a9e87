import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (camera-calibration_amd/), loaded under an importable name."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker only)."""
    import __graft_entry__ as ge
    return ge.load_oracle()


@pytest.fixture(scope="session")
def synth(pkg):
    return pkg.synth
