"""GPU parity of the posterior stage: the batched guide sampler and the dense E[log S] / E[log U]
summaries, both through the C ABI.

* vc_sample_posterior(n) must equal n single vc_sample_guide draws bit for bit (same Philox streams).
* vc_expected_logs must equal the reference's einsum formulas (velocity_inference_model.py:236-258,
  phase_inference_model.py:248-262), restated here op by op in torch float64; tolerance 2e-6 relative + 2e-6
  absolute (float32 sincos / log on the device).
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _engine(case):
    from velocycle_amd.engine import HipEngine
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    eng = HipEngine(spec)
    eng.set_params({k[4:]: torch.tensor(v) for k, v in z.items() if k.startswith("par_")})
    return eng, spec


def _sites(spec):
    s = ["ν", "ϕxy", "ϕ"]
    if spec.with_delta_nu:
        s.append("Δν")
    if spec.noisemodel == "NegativeBinomial":
        s.append("shape_inv")
    if spec.kind == "velocity":
        s += ["logγg", "logβg", "νω", "ω"]
        if spec.guide == "lrmn":
            s.append("rho_real")
    return s


@pytest.mark.parametrize("case", H.STEP_CASES)
def test_batched_draws_equal_single_draws(case):
    eng, spec = _engine(case)
    names = _sites(spec)
    n = 7
    out = eng.sample_posterior(names, n, seed=1234, step0=5)
    torch.cuda.synchronize()
    for i in range(n):
        eng.sample_guide(seed=1234, step=5 + i)
        for nm in names:
            one = eng.read_site(nm)
            got = out[nm][i].cpu().reshape(one.shape)
            assert torch.equal(got, one), f"{case}: draw {i} site {nm}"
    # the draws differ from each other wherever the guide is not a point mass / the site is not conditioned
    assert any(not torch.equal(out[nm][0], out[nm][1]) for nm in names)
    eng.close()


def test_batched_draws_reject_bad_sites():
    eng, spec = _engine("phase_nb" if "phase_nb" in H.STEP_CASES else H.STEP_CASES[0])
    if spec.kind == "phase":
        with pytest.raises(ValueError, match="not in this model"):
            eng.sample_posterior(["logγg"], 2)          # not a site of the phase model
    assert eng.sample_posterior(["ν"], 0)["ν"].shape[0] == 0     # zero draws: empty result, no launch error
    eng.close()


def _reference_summaries(spec, nu, dnu, phi, cf_avg, omega=None, logbeta=None, gamma=None):
    """velocity_inference_model.py:236-258 / phase_inference_model.py:248-262 in float64."""
    from velocycle_amd.utils import torch_fourier_basis
    f = lambda t: torch.as_tensor(t).double()
    ζ = torch_fourier_basis(f(phi), spec.H, der=0).double()
    base = torch.einsum("gh,ch->gc", f(nu).reshape(spec.Ng, spec.Nh), ζ)
    if spec.with_delta_nu:
        base = base + torch.einsum("bc,bg->gc", f(spec.Db).reshape(spec.Nb, spec.Nc), f(dnu).reshape(spec.Nb, spec.Ng))
    S = base + f(spec.count_factor).reshape(1, spec.Nc)
    S2 = base + cf_avg
    if omega is None:
        return S, S2
    ζd = torch_fourier_basis(f(phi), spec.H, der=1).double()
    dd = torch.einsum("gh,ch->gc", f(nu).reshape(spec.Ng, spec.Nh), ζd) * f(omega).reshape(1, -1)
    z = dd + f(gamma).reshape(-1, 1)
    core = -f(logbeta).reshape(-1, 1) + torch.log(torch.relu(z) + 1e-5)
    return S, S2, core + S, core + S2


def _tolerance(spec, nu, phi, omega, gamma, k, want):
    """2e-6 relative + 2e-6 absolute; for the ElogU outputs (k >= 2) plus the float32 rounding of the argument of
    log(relu(z) + 1e-5), which the log amplifies by 1 / (relu(z) + 1e-5) near the floor (the reference's own float32
    evaluation carries the same spread)."""
    tol = 2e-6 * want.abs() + 2e-6
    if k >= 2:
        from velocycle_amd.utils import torch_fourier_basis
        f = lambda t: torch.as_tensor(t).double()
        ζd = torch_fourier_basis(f(phi), spec.H, der=1).double().abs()
        mag = torch.einsum("gh,ch->gc", f(nu).reshape(spec.Ng, spec.Nh).abs(), ζd) * f(omega).abs().reshape(1, -1) \
            + f(gamma).reshape(-1, 1)
        dd = torch.einsum("gh,ch->gc", f(nu).reshape(spec.Ng, spec.Nh),
                          torch_fourier_basis(f(phi), spec.H, der=1).double()) * f(omega).reshape(1, -1)
        z = dd + f(gamma).reshape(-1, 1)
        tol = tol + 6e-7 * mag / (torch.relu(z) + 1e-5)
    return tol


@pytest.mark.parametrize("case", H.STEP_CASES)
def test_expected_logs_match_reference_formulas(case):
    eng, spec = _engine(case)
    g = torch.Generator().manual_seed(3)
    nu = torch.randn(spec.Ng, spec.Nh, generator=g) * 0.5
    phi = (torch.rand(spec.Nc, generator=g) * 2 - 1) * np.pi
    dnu = torch.randn(spec.Nb, spec.Ng, generator=g) * 0.1 if spec.with_delta_nu else None
    cf_avg = float(torch.as_tensor(spec.count_factor).float().mean())
    if spec.kind == "velocity":
        omega = torch.rand(spec.Nc, generator=g) * 0.6 - 0.1          # some negative speeds: the relu branch is hit
        logbeta = torch.randn(spec.Ng, generator=g) + 2.0
        gamma = torch.exp(torch.randn(spec.Ng, generator=g) * 0.5 - 1.0)
        got = eng.expected_logs(nu, phi, cf_avg, dnu=dnu, omega=omega, logbeta=logbeta, gamma=gamma)
        want = _reference_summaries(spec, nu, dnu, phi, cf_avg, omega, logbeta, gamma)
    else:
        got = eng.expected_logs(nu, phi, cf_avg, dnu=dnu)
        want = _reference_summaries(spec, nu, dnu, phi, cf_avg)
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for k, (a, b) in enumerate(zip(got, want)):
        a = a.cpu().double()
        assert a.shape == (spec.Ng, spec.Nc)
        err = (a - b).abs()
        tol = _tolerance(spec, nu, phi, omega if spec.kind == "velocity" else None,
                         gamma if spec.kind == "velocity" else None, k, b)
        assert bool((err <= tol).all()), f"{case}: output {k}: max excess {float((err - tol).max())}"
    eng.close()


def test_expected_logs_at_a_ragged_larger_size():
    """Nc not a multiple of 4 and several 1024-cell blocks: the vector-store / tail paths of the kernel."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(4099, 37, "vjoint", seed=2)
    eng = HipEngine(spec)
    g = torch.Generator().manual_seed(0)
    nu = torch.randn(spec.Ng, spec.Nh, generator=g)
    phi = (torch.rand(spec.Nc, generator=g) * 2 - 1) * np.pi
    omega = torch.full((spec.Nc,), 0.4)
    logbeta, gamma = torch.randn(spec.Ng, generator=g) + 2, torch.rand(spec.Ng, generator=g) + 0.1
    cf_avg = float(spec.count_factor.float().mean())
    got = eng.expected_logs(nu, phi, cf_avg, omega=omega, logbeta=logbeta, gamma=gamma)
    want = _reference_summaries(spec, nu, None, phi, cf_avg, omega, logbeta, gamma)
    for k, (a, b) in enumerate(zip(got, want)):
        err = (a.cpu().double() - b).abs()
        assert bool((err <= _tolerance(spec, nu, phi, omega, gamma, k, b)).all()), f"output {k}: {float(err.max())}"
    eng.close()


@pytest.mark.parametrize("case", H.FIT_CASES)
def test_expected_logs_match_the_reference_posterior(case):
    """vc_expected_logs against the numbers the REFERENCE's own posterior_sampling produced after its fit()
    (`fitm.posterior["ElogS" / "ElogS2" / "ElogU" / "ElogU2"]`, velocity_inference_model.py:236-262,
    phase_inference_model.py:248-265; stored by tests/golden/make_golden.py together with the fitted ν_locs, the
    fitted phases and the draw means γ, log β, νω they were computed from)."""
    from velocycle_amd.engine import HipEngine
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_{case}.npz")
    spec = H.spec_from_fixture(z)
    eng = HipEngine(spec)
    nu = torch.tensor(z["reffit_ν_locs"])
    phis = torch.tensor(z["post_phis"])
    dnu = torch.tensor(z["reffit_Δν_locs"]) if spec.with_delta_nu else None
    kw = {}
    if spec.kind == "velocity":
        nuw = torch.tensor(z["post_nuw_mean"]).double()                      # (Nx, Nhw)
        p = phis.double()
        zw = torch.stack([torch.ones_like(p)] + [f((k + 1) * p) for k in range(spec.Hw) for f in (torch.sin, torch.cos)])
        omega = ((nuw @ zw) * spec.D.double()).sum(0)                        # one speed per cell (N2)
        kw = dict(omega=omega.float(), logbeta=torch.tensor(z["post_logbeta_mean"]), gamma=torch.tensor(z["post_gamma_mean"]))
        # the reference's draws of the deterministic site ω are that same contraction at each draw's νω and ϕ
        pd, nd = torch.tensor(z["post_phi_draws"]).double(), torch.tensor(z["post_nuw_draws"]).double()
        zd = torch.stack([torch.ones_like(pd)] + [f((k + 1) * pd) for k in range(spec.Hw) for f in (torch.sin, torch.cos)], 1)
        om_d = torch.einsum("nxh,nhc,xc->nc", nd, zd, spec.D.double())
        assert np.allclose(om_d.numpy(), z["post_omega_draws"], rtol=1e-5, atol=1e-6)
    outs = eng.expected_logs(nu, phis, float(z["post_cf_avg"]), dnu=dnu, **kw)
    torch.cuda.synchronize()
    names = ["ElogS", "ElogS2"] + (["ElogU", "ElogU2"] if spec.kind == "velocity" else [])
    for n, o in zip(names, outs):
        want = z["post_" + n]
        assert np.allclose(o.cpu().numpy(), want, rtol=2e-5, atol=2e-5), (n, np.abs(o.cpu().numpy() - want).max())
    eng.close()
