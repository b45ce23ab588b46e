"""GPU: the drop-in Python surface end to end -- preprocess_for_* -> {Phase,Velocity}FitModel.fit -> attributes
-- against what the reference's own fit() produced on the same inputs and seed (tests/golden/ref_fit_*.npz)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _objects(z):
    from velocycle_amd import containers as C
    from velocycle_amd.anndata_lite import AnnDataLite
    S, U = z["in_S"].T, (z["in_U"].T if "in_U" in z else z["in_S"].T * 0)
    ad = AnnDataLite(S, U)
    genes = list(ad.var.index)
    cyc = C.Cycle.from_array(z["in_mu_nu"].T, z["in_sd_nu"].T, genes)
    ph = C.Phases.from_array(z["in_phixy_prior"].T, cell_names=list(ad.obs.index))
    Db = torch.tensor(z["in_Db"].T)
    return ad, cyc, ph, Db


def _opt(z):
    from velocycle_amd.optim import ClippedAdam
    return ClippedAdam({"lr": float(z["opt_lr"]), "lrd": float(z["opt_lrd"]),
                        "betas": tuple(float(x) for x in z["opt_betas"])})


def _close(a, b, tol=2e-3):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.allclose(a, b, rtol=tol, atol=tol), np.abs(a - b).max()


def test_phase_fit_drop_in():
    from velocycle_amd import preprocessing as P
    from velocycle_amd.fit_models import PhaseFitModel
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_phase_nb.npz")
    ad, cyc, ph, Db = _objects(z)
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=int(z["in_H"]), with_delta_nu=False)
    PhaseFitModel._default_elbo_fresh = True
    fit = PhaseFitModel(mp, num_samples=6, n_per_bin=3)
    fit.fit(_opt(z), num_steps=int(z["num_steps"]), verbose=False, mode="parity", seed=int(z["seed"]))
    assert np.allclose(fit.losses, z["ref_losses"], rtol=1e-4, atol=1e-2)
    for attr in ("phis_pyro", "fourier_coef", "fourier_coef_sd", "disp_pyro"):
        _close(getattr(fit, attr), z["attr_" + attr])
    assert fit.cycle_pyro.means.shape == (3, mp.Ng) and len(fit.phase_pyro) == mp.Nc
    post = fit.posterior
    assert post["ν"].shape == (6, mp.Ng, 1, 3) and post["ϕxy"].shape == (6, mp.Nc, 2)
    assert post["ζ"].shape == (6, mp.Nc, 3) and post["shape_inv"].shape == (6, mp.Ng, 1)
    assert post["ElogS"].shape == (mp.Ng, mp.Nc) and torch.isfinite(post["ElogS2"]).all()
    # the deterministic posterior summaries against the reference's own posterior (phase_inference_model.py:248-265)
    _close(post["ElogS"], z["post_ElogS"], 5e-3)
    _close(post["ElogS2"], z["post_ElogS2"], 5e-3)
    # posterior draws scatter around the fitted means with the fitted scales
    assert np.allclose(post["ϕxy"].mean(0).numpy(), fit.phis_pyro.T, atol=2.0)
    # a second fit of the same class skips Trace_ELBO's warm-up pass, like the reference's shared loss object
    assert PhaseFitModel._default_elbo_fresh is False


@pytest.mark.parametrize("case", ["vel_lrmn_cond", "vel_mf_joint", "vel_mf_cond"])
def test_velocity_fit_drop_in(case):
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.fit_models import VelocityFitModel
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_{case}.npz")
    ad, cyc, ph, Db = _objects(z)
    Hw = int(z["in_Hw"])
    spd = C.AngularSpeed.from_array(z["in_mu_nuw"].T if Hw else z["in_mu_nuw"].reshape(-1),
                                    z["in_sd_nuw"].T if Hw else z["in_sd_nuw"].reshape(-1), ["b0"], Nhω=2 * Hw + 1)
    cond = {}
    for k, v in z.items():
        if k.startswith("cond_"):
            name = k[5:]
            t = torch.tensor(v)
            cond[name] = {"ν": lambda t: t.unsqueeze(-2), "shape_inv": lambda t: t.unsqueeze(-1)}.get(name, lambda t: t)(t)
    mp = P.preprocess_for_velocity_estimation(
        ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=int(z["in_H"]), ω_n_harmonics=Hw,
        count_factor=torch.tensor(z["in_count_factor"])[None, None, :], with_delta_nu=False, condition_on=cond,
        model_type="lrmn" if str(z["in_guide"]) == "lrmn" else "normal")
    VelocityFitModel._default_elbo_fresh = True
    fit = VelocityFitModel(mp, condition_on=cond, num_samples=6, n_per_bin=3)
    fit.fit(_opt(z), num_steps=int(z["num_steps"]), verbose=False, mode="parity", seed=int(z["seed"]))
    assert np.allclose(fit.losses, z["ref_losses"], rtol=1e-4, atol=1e-2)
    for attr in ("phis_pyro", "fourier_coef", "fourier_coef_sd", "disp_pyro", "log_betas"):
        _close(getattr(fit, attr), z["attr_" + attr])
    if str(z["in_guide"]) != "lrmn":
        for attr in ("log_gammas", "velocity_coef", "velocity_coef_sd"):
            _close(getattr(fit, attr), z["attr_" + attr])
    else:   # LRMN: log_gammas / speed come from posterior draws (different RNG stream): check against fitted `loc`
        loc = fit.engine.named()["loc"].cpu().numpy()
        assert np.abs(fit.log_gammas - loc[: mp.Ng]).max() < 2.0
        assert fit.speed_pyro.means.shape == (2 * Hw + 1, 1)
    assert fit.posterior["ElogU"].shape == (mp.Ng, mp.Nc) and fit.posterior["ω"].shape == (6, 1, mp.Nc)
    # ElogS / ElogS2 depend on the fitted ν_locs and phases only: equal to the reference's posterior
    # (velocity_inference_model.py:236-247); ElogU additionally on the draw means (own RNG stream) -> tests/test_hip_posterior.py
    _close(fit.posterior["ElogS"], z["post_ElogS"], 5e-3)
    _close(fit.posterior["ElogS2"], z["post_ElogS2"], 5e-3)
    assert fit.speed_pyro.conditions == ["b0"]


def test_perf_mode_fit_recovers_structure():
    """mode="perf" (Philox eps, hipGraph): loss decreases, phases stay correlated with the truth."""
    from velocycle_amd.fit_models import run_svi
    from velocycle_amd.workloads import make_velocity_spec
    from velocycle_amd.utils import circular_corrcoef
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.anndata_lite import AnnDataLite
    sp = make_velocity_spec(2000, 120, "vjoint", seed=3)
    ad = AnnDataLite(sp.S.t().numpy(), sp.U.t().numpy())
    cyc = C.Cycle.from_array(sp.mu_nu.T.numpy(), sp.sd_nu.T.numpy(), list(ad.var.index))
    ph = C.Phases.from_array(sp.phixy_prior.T.numpy(), cell_names=list(ad.obs.index))
    spd = C.AngularSpeed.trivial_prior(["c0"], harmonics=1)
    D = torch.ones(2000, 1)
    mp = P.preprocess_for_velocity_estimation(ad, cyc, ph, spd, D, D, n_harmonics=1, ω_n_harmonics=1,
                                              count_factor=sp.count_factor[None, None, :], with_delta_nu=False,
                                              model_type="normal")
    torch.manual_seed(0)
    fit = run_svi(mp, {"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)}, num_steps=300)
    assert len(fit.losses) == 300 and fit.losses[-1] < fit.losses[0]
    est = np.arctan2(fit.phis_pyro[1], fit.phis_pyro[0])
    assert circular_corrcoef(est, sp.truth["phis"].cpu().numpy()) > 0.8


def test_early_exit_and_store_output_follow_the_reference_loop(monkeypatch):
    """early_exit: armed after step 200, then checked every step (|mean(last 100) - mean(last 10)| < 5 -> break),
    exactly as velocity_inference_model.py:147-151; store_output returns posterior snapshots."""
    from velocycle_amd import preprocessing as P
    from velocycle_amd.fit_models import PhaseFitModel
    from velocycle_amd import svi
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_phase_nb.npz")
    ad, cyc, ph, Db = _objects(z)
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=int(z["in_H"]), with_delta_nu=False)
    calls = {"n": 0}
    real_step = svi.SVIRunner.step

    def flat_step(self, eps=None):
        real_step(self, eps)
        calls["n"] += 1
        return 1000.0 + (calls["n"] % 2)          # a converged, flat loss curve
    monkeypatch.setattr(svi.SVIRunner, "step", flat_step)
    fit = PhaseFitModel(mp, early_exit=True, num_samples=2, n_per_bin=2)
    fit.fit(_opt(z), num_steps=400, verbose=False, mode="parity", seed=1)
    assert len(fit.losses) == 203                 # steps 0..201 arm the check, step 202 breaks
    fit2 = PhaseFitModel(mp, num_samples=2, n_per_bin=2)
    out = fit2.fit(_opt(z), num_steps=5, verbose=False, mode="parity", seed=1, store_output=True,
                   intermediate_output_step_size=2)
    assert len(out) == 3 and out[0]["ν"].shape[0] == 2 and len(fit2.losses) == 5


def test_two_stage_tutorial_flow_matches_reference():
    """Phase fit -> hand-over built exactly like tutorial cell 42 -> default (LRMN) two-condition velocity fit,
    against the same flow run with the reference's own classes (tests/golden/ref_tutorial_flow.npz)."""
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.fit_models import PhaseFitModel, VelocityFitModel
    from velocycle_amd.optim import ClippedAdam
    from velocycle_amd import pyro_compat as pyro
    z = H.load_fixture(f"{H.GOLDEN}/ref_tutorial_flow.npz")
    ad = AnnDataLite(z["S"], z["U"])
    ad.obs["batch"] = list(z["batch"])
    cyc = C.Cycle.trivial_prior(list(ad.var.index), harmonics=1)
    cyc.set_means(z["cyc_means"])
    cyc.set_stds(z["cyc_stds"])
    ph = C.Phases.from_array(z["phi_xy"], cell_names=list(ad.obs.index))
    Db = P.make_design_matrix(ad, ids="batch")
    n1, n2, seed = int(z["n1"]), int(z["n2"]), int(z["seed"])
    opt = lambda n: ClippedAdam({"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)})
    pyro.clear_param_store()
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1, σΔν=torch.tensor(z["sd_dnu"]))
    PhaseFitModel._default_elbo_fresh = True
    pf = PhaseFitModel(mp, num_samples=4, n_per_bin=2)
    pf.fit(opt(n1), num_steps=n1, verbose=False, mode="parity", seed=seed)
    assert np.allclose(pf.losses, z["phase_losses"], rtol=1e-4, atol=1e-2)
    for a in ("phis_pyro", "fourier_coef", "fourier_coef_sd", "disp_pyro", "delta_nus"):
        _close(getattr(pf, a), z["phase_" + a])
    cond = {"ϕxy": pf.phase_pyro.phi_xy_tensor.T, "ν": pf.cycle_pyro.means_tensor.T.unsqueeze(-2),
            "Δν": torch.tensor(pf.delta_nus), "shape_inv": torch.tensor(pf.disp_pyro).unsqueeze(-1)}
    spd = C.AngularSpeed.trivial_prior(condition_names=["b0", "b1"], harmonics=0)
    pyro.clear_param_store()
    mv = P.preprocess_for_velocity_estimation(ad, pf.cycle_pyro, pf.phase_pyro, spd, Db.float(), Db.float(), n_harmonics=1,
                                              count_factor=mp.count_factor, ω_n_harmonics=0, condition_on=cond)
    assert mv.model_type == "lrmn" and mv.Nx == 2
    VelocityFitModel._default_elbo_fresh = True
    vf = VelocityFitModel(mv, condition_on=cond, num_samples=4, n_per_bin=2)
    vf.fit(opt(n2), num_steps=n2, verbose=False, mode="parity", seed=seed + 1)
    assert "vu_" in vf.engine.stats["main_kernel"]            # S term hoisted in the tutorial flow
    assert np.allclose(vf.losses, z["vel_losses"], rtol=1e-4, atol=1e-2)
    for a in ("phis_pyro", "fourier_coef", "disp_pyro", "log_betas", "delta_nus"):
        _close(getattr(vf, a), z["vel_" + a])
    _close(pyro.param("loc").numpy(), z["vel_loc"])
    _close(pyro.param("logβg_scales").numpy(), z["vel_logβg_scales"])
    assert vf.speed_pyro.means.shape == (1, 2) and list(vf.speed_pyro.conditions) == ["b0", "b1"]
