import os
import sys

import pytest

# torch ships its own copy of the HIP runtime.  Imported before libccalib.so is loaded, the library binds to that
# copy (same SONAME) and the process holds ONE runtime; loaded the other way round the process holds two, and a
# stream created by one (torch.cuda.Stream) is an invalid argument to the other.  Tests that hand torch streams
# or tensors to the library need the single-runtime order, so it is fixed here for the whole session.
import torch  # noqa: F401,E402

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
