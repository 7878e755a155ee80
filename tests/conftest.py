import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure, oracle/psm_oracle.c) - built on demand with gcc."""
    from oracle import psm_oracle_py as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def hip_lib():
    """libprimesm_hip.so - must exist (built by __graft_entry__.build() / make); no fallback."""
    from primestereomatch_amd import capi
    return capi.load()
