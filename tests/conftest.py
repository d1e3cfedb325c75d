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


def pytest_sessionstart(session):
    """A fresh checkout has no built library (it is git-ignored): build it in-tree once, if hipcc is here.
    (hipcc cross-compiles without a GPU; on the GPU box the prebuilt .so travels with the snapshot.)"""
    from open_headstage_amd import build as _build
    if not os.path.exists(_build.LIB):
        try:
            _build.build()
        except Exception as e:      # noqa: BLE001 -- the tests that need the library will say so themselves
            print(f"conftest: could not build {_build.LIB}: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def oracle():
    from oracle import ohs_oracle
    ohs_oracle.build()
    return ohs_oracle
