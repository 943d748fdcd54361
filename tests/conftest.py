import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu under gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libpfa_hip.so is the product; build it in-tree if it is missing or stale
    (hipcc cross-compiles gfx950 without a GPU).  The C oracle is built the same way."""
    from polyfasta_amd import _build
    _build.build()
    from oracle import c_oracle
    c_oracle.build()
    yield


@pytest.fixture(scope="session")
def gpu():
    from polyfasta_amd import _lib
    _lib.init(0)
    yield _lib
