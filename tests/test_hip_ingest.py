"""GPU: device-side ingest (reference: the `.A` densification of preprocessing.py:141-147, 243-252).

* the per-gene count histograms built on the device during the re-layout are IDENTICAL to the host pass (VC_HOST_HIST=1
  keeps the host pass as the checker), on fixtures, at a medium size with large / non-integer counts (overflow list), and
  at BASELINE's 50k x 2k;
* CSR input (what AnnData layers hold) gives the same engine -- histograms, loss, every gradient, bit for bit -- as the
  dense input, without the dense matrix ever being formed on the host or uploaded;
* invalid counts (negative / NaN / Inf) are refused with VC_ERR_ARG instead of poisoning the histograms."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _mk(spec, **kw):
    from velocycle_amd.engine import HipEngine
    return HipEngine(spec, **kw)


def _hist_equal(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b))


def _host_checker(spec, tuning=None, **kw):
    from velocycle_amd.tuning import Tuning
    e = _mk(spec, tuning=(tuning or Tuning()).replace(host_hist=True), **kw)
    assert not e.stats["hist_on_device"]
    return e


@pytest.mark.parametrize("case", ["vel_mf_joint", "phase_nb", "vel_lrmn_cond_dnu2", "vel_mf_poisson"])
def test_device_histograms_equal_host_pass_on_fixtures(case):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    # like with like: the (value, multiplicity) lists on both sides (the S+U models evaluate the histogram sums from dense
    # tail-count tables by default since round 4 -- another formula for the same numbers: checked at the end)
    from velocycle_amd.tuning import Tuning
    lists = Tuning(hist_dense="lists")
    dev, host = _mk(spec, tuning=lists), _host_checker(spec, lists)
    assert dev.stats["hist_on_device"]
    assert _hist_equal(dev.histogram(), host.histogram())
    for e in (dev, host):
        e.init_params() if spec.guide != "lrmn" else e.init_params(torch.zeros(spec.Ng + spec.Nx * spec.Nhw, spec.rho_rank))
        e.elbo_grad(eps=None, seed=3, step=1)
    torch.cuda.synchronize()
    # the constant sum lgamma(k+1) is added up per gene on one side, per host thread on the other: last-digit fp64 rounding
    assert abs(dev.loss() - host.loss()) <= 1e-13 * abs(host.loss())
    assert torch.equal(torch.nan_to_num(dev.grad[4:]), torch.nan_to_num(host.grad[4:]))
    # dense tail-count tables (sum_j C_j log(r + j), hardware log2 / reciprocal) against the lists (Stirling differences): the
    # same sums to float32 rounding of their terms
    dense = _mk(spec, tuning=Tuning(hist_dense="dense"))
    dense.init_params() if spec.guide != "lrmn" else dense.init_params(torch.zeros(spec.Ng + spec.Nx * spec.Nhw, spec.rho_rank))
    dense.elbo_grad(eps=None, seed=3, step=1)
    torch.cuda.synchronize()
    assert abs(dense.loss() - host.loss()) <= 2e-6 * abs(host.loss())
    gd, gh = torch.nan_to_num(dense.grad[4:]).double(), torch.nan_to_num(host.grad[4:]).double()
    assert float((gd - gh).abs().max()) <= 1e-4 * float(gh.abs().max())
    dev.close(); host.close(); dense.close()


def _spiky_spec(on_device):
    """3001 x 300 with a few huge counts (>= the dense-bin cap) and non-integer values: exercises the overflow list."""
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3001, 300, "vjoint", 1, 1, seed=9)
    S, U = spec.S.contiguous().clone(), spec.U.contiguous().clone()
    g = torch.Generator().manual_seed(1)
    for M in (S, U):
        idx = torch.randint(0, M.numel(), (500,), generator=g)
        M.view(-1)[idx[:250]] = torch.randint(2048, 70000, (250,), generator=g).float()
        M.view(-1)[idx[250:]] = torch.rand(250, generator=g) * 30 + 0.25
    spec.S, spec.U = (S.cuda(), U.cuda()) if on_device else (S, U)
    return spec


@pytest.mark.parametrize("on_device", [False, True])
def test_device_histograms_with_overflow_values(on_device):
    spec = _spiky_spec(on_device)
    dev, host = _mk(spec), _host_checker(spec)
    hd, hh = dev.histogram(), host.histogram()
    assert _hist_equal(hd, hh) and hd[1].max() >= 2048 and (hd[1] != np.floor(hd[1])).any()
    # two shards: each rank's histograms come from its own cells only
    a, b = _mk(spec, rank=1, world_size=2), _host_checker(spec, rank=1, world_size=2)
    assert _hist_equal(a.histogram(), b.histogram()) and not _hist_equal(a.histogram(), hd)
    for e in (dev, host, a, b):
        e.close()


def test_device_histograms_full_size():
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(50000, 2000, "vjoint", 1, 1, seed=0, device="cuda")
    dev = _mk(spec)
    hd = dev.histogram()
    transient = dev.stats["setup_transient_bytes"]
    dev.close()
    host = _host_checker(spec)
    assert _hist_equal(hd, host.histogram())
    # histogram tables + overflow lists (+ the float32 layout while its uint16 copy is made): never a copy of the matrix
    # in the caller's layout, never a byte of it back on the host
    blocked_f32 = 2 * 4 * 2048 * 50000
    assert transient <= 2 * (2000 * 2048 * 4 + (1 << 22) * 8) + 64 + blocked_f32
    host.close()


@pytest.mark.parametrize("mode,ncond", [("vjoint", 2), ("vcond", 1)])
def test_csr_ingest_equals_dense_ingest(mode, ncond):
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(2003, 260, mode, ncond, 1, seed=4)
    dense = _mk(spec)
    import copy
    spec2 = copy.copy(spec)
    spec2.S_csr = sp.csr_matrix(spec.S.t().numpy())
    spec2.U_csr = sp.csr_matrix(spec.U.t().numpy())
    spec2.S = None
    spec2.U = None
    csr = _mk(spec2)
    assert spec2.Ng == spec.Ng and spec2.Nc == spec.Nc
    assert _hist_equal(dense.histogram(), csr.histogram())
    g = torch.Generator().manual_seed(0)
    first = draw_eps(spec, g)
    eps = draw_eps(spec, g)
    for e in (dense, csr):
        e.init_params(first.get("_cov_factor_draw"))
        e.elbo_grad(eps=e.pack_eps(eps))
    torch.cuda.synchronize()
    assert abs(dense.loss() - csr.loss()) <= 1e-13 * abs(csr.loss())
    assert torch.equal(torch.nan_to_num(dense.grad[4:]), torch.nan_to_num(csr.grad[4:]))
    # sharded CSR: rank 1 of 3
    a, b = _mk(spec, rank=1, world_size=3), _mk(spec2, rank=1, world_size=3)
    for e in (a, b):
        e.init_params(first.get("_cov_factor_draw"))
        e.elbo_grad(eps=e.pack_eps(eps))
    torch.cuda.synchronize()
    assert abs(a.loss() - b.loss()) <= 1e-13 * abs(b.loss()) and torch.equal(torch.nan_to_num(a.grad[4:]), torch.nan_to_num(b.grad[4:]))
    for e in (dense, csr, a, b):
        e.close()


def test_sparse_anndata_layers_reach_the_engine_as_csr():
    """preprocess_for_* keep scipy-sparse layers as S_csr / U_csr next to the dense S / U of the reference's contract, and
    the fit drivers hand them to the engine; result identical to the dense-layer fit."""
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.fit_models import PhaseFitModel
    from velocycle_amd.workloads import make_phase_spec
    spec = make_phase_spec(900, 80, seed=2)
    S, U = spec.S.t().numpy(), spec.S.t().numpy() * 0
    out = []
    from velocycle_amd import pyro_compat as pyro
    for sparse in (False, True):
        pyro.clear_param_store()          # (a fit() continues from the store otherwise: the second would start where the first ended)
        ad = AnnDataLite(sp.csr_matrix(S) if sparse else S, sp.csr_matrix(U) if sparse else U)
        cyc = C.Cycle.from_array(spec.mu_nu.T.numpy(), spec.sd_nu.T.numpy(), list(ad.var.index))
        ph = C.Phases.from_array(spec.phixy_prior.T.numpy(), cell_names=list(ad.obs.index))
        mp = P.preprocess_for_phase_estimation(ad, cyc, ph, torch.ones(900, 1), n_harmonics=1, with_delta_nu=False)
        assert (mp.S_csr is not None) == sparse
        fit = PhaseFitModel(mp, num_samples=2, n_per_bin=2)
        fit.fit({"lr": 0.03, "lrd": 0.99, "betas": (0.8, 0.99)}, num_steps=20, verbose=False, seed=5)
        out.append((np.array(fit.losses), fit.fourier_coef.copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("bad", [-1.0, float("nan"), float("inf")])
def test_invalid_counts_are_refused(bad):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_vel_mf_joint.npz")
    spec = H.spec_from_fixture(z)
    spec.U = spec.U.clone()
    spec.U[3, 5] = bad
    with pytest.raises(ValueError, match="finite and >= 0"):
        _mk(spec)


@pytest.mark.parametrize("mode", ["vjoint", "vcond", "phase"])
def test_uint16_count_storage_equals_float32(mode):
    """Counts that are integers <= 65535 are stored as uint16 in HBM (half the bytes K_main streams); Tuning(count_storage="f32")
    keeps the reference's float32.  Same arithmetic on the same values: loss and every gradient agree to float32 rounding
    of re-scheduled FMAs (observed bit-identical); a matrix with one count > 65535 falls back to float32 by itself."""
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    spec = make_phase_spec(2500, 300, seed=3) if mode == "phase" else make_velocity_spec(2500, 300, mode, 2, 1, seed=3)
    g = torch.Generator().manual_seed(0)
    first = draw_eps(spec, g)
    eps = draw_eps(spec, g)
    res = {}
    from velocycle_amd.tuning import Tuning
    for storage in ("u16", "f32"):
        e = _mk(spec, tuning=Tuning(count_storage="f32" if storage == "f32" else None))
        assert e.stats["count_storage"] == storage and ("u16" in e.stats["main_kernel"]) == (storage == "u16")
        e.init_params(first.get("_cov_factor_draw"))
        e.elbo_grad(eps=e.pack_eps(eps))
        torch.cuda.synchronize()
        res[storage] = (e.loss(), e.grad.clone().cpu(), e.stats["streamed_bytes"])
        e.close()
    assert res["u16"][2] * 2 == res["f32"][2]
    assert abs(res["u16"][0] - res["f32"][0]) <= 1e-9 * abs(res["f32"][0])
    a, b = torch.nan_to_num(res["u16"][1][4:]).double(), torch.nan_to_num(res["f32"][1][4:]).double()
    assert float((a - b).abs().max()) <= 1e-6 * max(float(b.abs().max()), 1.0)
    # one huge count: the whole rank falls back to float32 storage, results unchanged in kind
    import copy
    big = copy.copy(spec)
    big.S = spec.S.contiguous().clone()
    big.S[1, 7] = 70000.0
    e = _mk(big)
    assert e.stats["count_storage"] == "f32"
    e.close()
