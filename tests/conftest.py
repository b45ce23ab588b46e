import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fresh_param_store():
    """The param store is process-global, as Pyro's is: a fit() continues from whatever an earlier fit left there
    (velocycle_amd/pyro_compat.py).  Tests are independent of each other: each starts with an empty store."""
    try:
        from velocycle_amd import pyro_compat
        pyro_compat.clear_param_store()
    except Exception:
        pass
    yield
