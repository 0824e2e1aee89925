import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / at round end)")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    return oracle_lib.oracle()


@pytest.fixture(scope="session")
def emu():
    import oracle_lib
    return oracle_lib.emul()


@pytest.fixture(scope="session")
def hal():
    """One HipHal context for the whole GPU test session (one process on the card)."""
    from raiko_amd.hal import HipHal
    h = HipHal(0)
    yield h
    h.close()
