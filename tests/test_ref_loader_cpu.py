"""CPU: which Pyro the fixture generators run the reference on (oracle/ref_loader.py; VERDICT r5 missing #2).

pyro-ppl==1.8.6 (reference requirements.txt:105) is not installable in the build container: every committed fixture comes from
oracle/pyro_shim.  The switch must (a) fall back to the shim when the library is absent, (b) refuse `--real-pyro` then, (c) pick the
real library the day it is importable -- proved here with a stand-in distribution that only carries the version metadata --, and
(d) `make_golden.py --check` must reproduce the committed fixtures from the generator (needs /root/reference: build container)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader  # noqa: E402


def test_without_pyro_the_switch_falls_back_to_the_shim():
    if ref_loader.real_pyro_version() is not None:
        pytest.skip("a real pyro-ppl is installed here")
    assert ref_loader.choose_backend("auto") == ("shim", None)
    assert ref_loader.choose_backend("shim") == ("shim", None)
    with pytest.raises(RuntimeError, match="real-pyro"):
        ref_loader.choose_backend("real")
    with pytest.raises(ValueError):
        ref_loader.choose_backend("other")


def _fake_dist(tmp_path, version):
    (tmp_path / "pyro").mkdir()
    (tmp_path / "pyro" / "__init__.py").write_text(f"__version__ = '{version}'\n")
    di = tmp_path / f"pyro_ppl-{version}.dist-info"
    di.mkdir()
    (di / "METADATA").write_text(f"Metadata-Version: 2.1\nName: pyro-ppl\nVersion: {version}\n")
    return str(tmp_path)


def test_an_installed_pyro_of_the_pinned_series_is_chosen(tmp_path, monkeypatch):
    if ref_loader.real_pyro_version() is not None:
        pytest.skip("a real pyro-ppl is installed here")
    monkeypatch.syspath_prepend(_fake_dist(tmp_path, "1.8.6"))
    assert ref_loader.real_pyro_version() == "1.8.6"
    assert ref_loader.choose_backend("auto") == ("real", "1.8.6")
    assert ref_loader.choose_backend("real") == ("real", "1.8.6")
    assert ref_loader.choose_backend("shim") == ("shim", None)


def test_another_series_is_not_mistaken_for_the_pinned_one(tmp_path, monkeypatch):
    if ref_loader.real_pyro_version() is not None:
        pytest.skip("a real pyro-ppl is installed here")
    monkeypatch.syspath_prepend(_fake_dist(tmp_path, "1.9.1"))
    assert ref_loader.choose_backend("auto") == ("shim", None)
    with pytest.raises(RuntimeError, match="1.9.1"):
        ref_loader.choose_backend("real")


def test_the_shim_is_never_taken_for_the_library(monkeypatch):
    monkeypatch.syspath_prepend(ref_loader.SHIM)
    if ref_loader.real_pyro_version() is None:
        assert ref_loader.choose_backend("auto") == ("shim", None)


def test_the_param_store_is_read_the_way_the_library_exposes_it():
    """`named_parameters()` (pyro-ppl 1.8.6 ParamStoreDict) is the one accessor the generators use: the shim has it."""
    sys.path.insert(0, ref_loader.SHIM)
    try:
        import importlib
        rt = importlib.import_module("pyro.runtime")
        import torch
        from torch.distributions import constraints
        st = rt.ParamStore()
        st.setdefault("a", torch.tensor([2.0]), constraints.positive)
        u = ref_loader.unconstrained_params(st)
        assert set(u) == {"a"} and abs(float(u["a"].detach()) - 0.6931472) < 1e-6 and u["a"].requires_grad
    finally:
        sys.path.remove(ref_loader.SHIM)
        for m in [m for m in sys.modules if m == "pyro" or m.startswith("pyro.")]:
            del sys.modules[m]


@pytest.mark.skipif(not ref_loader.reference_available(), reason="needs /root/reference (build container only)")
def test_check_mode_reproduces_the_committed_fixtures():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check", "phase_nb", "vel_lrmn_cond"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "every regenerated array agrees" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check", "--real-pyro", "phase_nb"],
                       capture_output=True, text=True, timeout=600)
    if ref_loader.real_pyro_version() is None:
        assert r.returncode != 0 and "real-pyro" in r.stderr
