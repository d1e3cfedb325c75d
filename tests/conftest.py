"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` runs on a CPU-only container (oracle vs golden vectors, host logic,
C-ABI symbol checks).  `-m gpu` runs on a real MI355X and calls the HIP path
through the C-ABI shared library.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import ohs_oracle
    ohs_oracle.build()
    return ohs_oracle
