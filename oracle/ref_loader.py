"""Import the reference package (unmodified, from /root/reference/build/lib) on top of the pyro shim.

TEST INFRASTRUCTURE ONLY -- works only in the build container (the GPU box has no /root/reference).
Used by tests/golden/make_golden.py and by tests that are skipped when the reference is absent.
"""
import importlib
import os
import sys

REF_LIB = "/root/reference/build/lib"
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyro_shim")


def reference_available():
    return os.path.isdir(os.path.join(REF_LIB, "velocycle"))


def load_reference():
    """Returns the imported reference `velocycle` package (its own code; `pyro` is the shim)."""
    if not reference_available():
        raise RuntimeError("reference tree not present")
    if SHIM not in sys.path:
        sys.path.insert(0, SHIM)
    if REF_LIB not in sys.path:
        sys.path.insert(1, REF_LIB)
    import matplotlib
    matplotlib.use("Agg")
    return importlib.import_module("velocycle")
