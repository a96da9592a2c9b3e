import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Load the solver library (which initialises its HIP runtime) before any test imports
# a package that carries a HIP runtime of its own.
if os.path.exists(os.path.join(ROOT, "krylov_amd", "libkrylov_amd.so")):
    try:
        from krylov_amd import capi as _capi
        _capi.lib()
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle
