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


def pytest_terminal_summary(terminalreporter):
    """How often the gradient-block check of tests/helpers.py passed only through its `4 x float32 oracle` clause."""
    try:
        from tests import helpers as H
    except Exception:
        return
    st = H.CLAUSE_STATS
    if st["blocks"]:
        terminalreporter.write_line(f"assert_step_matches_oracle: {st['blocks']} gradient blocks checked, "
                                    f"{len(st['by_ref32_clause'])} passed through the 4 x float32-oracle clause only"
                                    + ("" if not st["by_ref32_clause"] else ": " + ", ".join(f"{n} (err {e:.1e} / float32 oracle {r:.1e} of the block max)" for n, e, r in st["by_ref32_clause"][:12])))
