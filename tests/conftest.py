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
    """The CPU oracle (test infrastructure): built on demand with g++."""
    from oracle import oracle as O
    O.build()
    return O


_SWITCHES = ("ASURF_DEBUG", "ASURF_WALK_SEG", "ASURF_WALK", "ASURF_RPW", "ASURF_NO_PIPELINE", "ASURF_CUVOL_RPW",
             "ASURF_NO_REPLAY", "ASURF_NO_SPLIT_BWD", "ASURF_EVAL", "ASURF_MASKS", "ASURF_SWEEP")


@pytest.fixture(autouse=True)
def _reset_library_switches():
    """The internal switches a test sets through lib.config_set() do not outlive it (a failing test would otherwise
    leave the library in its configuration for every test that follows).  Only when the library is loaded already."""
    yield
    from alphasurf_amd import lib
    if lib._lib is not None:
        for name in _SWITCHES:
            lib.config_set(name, None)
