"""GPU acceptance tests of BASELINE.json's north_star criterion on REAL-LENGTH fits (VERDICT r2 "missing" #1):
"posterior means within 1e-3 rel of reference" and SURVEY §4(d) parameter recovery ("omega recovered within CI").

1. Parity mode (host eps in the reference's RNG order) for 1 500 steps with the tutorials' learning-rate schedule against the
   float64 oracle's trajectory on the same eps stream (tests/golden/oracle_fit_*.npz, written by
   tests/golden/make_oracle_fits.py from oracle.fit): the posterior means (ν_locs, logγ / loc[:Ng], logβg_locs, νω, ϕxy_locs)
   and scales within 1e-3 of each block's max-norm -- or 4x the float32 oracle's own distance from float64 where Adam has
   amplified rounding (float32 is what the reference computes in).  Reference loop: velocity_inference_model.py:118-187.
2. Performance mode (the benchmarked fused3 / Philox path): converged fits return the angular speed the REFERENCE'S MODEL
   returns on the same data -- the float64 oracle's own converged fit -- within the posterior's width, the simulated
   phases, and the simulated RATIO of the speeds of two samples (0.3 / 0.4; utils.py:508, 539-543 is the recipe).
   The absolute speed is not what was simulated, for the oracle either: on these workloads (priors set the way the
   tutorials set them) the model's posterior mean of ω sits 20-35 % above the simulated value in every implementation
   (measured: oracle 0.540 / HIP 0.530 for ω = 0.4; the γ prior fixes the time scale, and the prior on the harmonics of ν
   shrinks their amplitude).  The tests therefore hold the HIP fit to the oracle's, and to the simulation in what the
   model identifies (ratio, phases, order of magnitude)."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H
from tests.golden import make_oracle_fits as G

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.parametrize("name", sorted(G.CASES))
def test_converged_parity_fit_matches_the_fp64_oracle_trajectory(name):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    z = np.load(os.path.join(H.GOLDEN, f"oracle_fit_{name}.npz"))
    spec = G.make_spec(name)
    assert G.digest(spec) == str(z["digest"]), "the synthetic workload was not rebuilt bit for bit"
    n, seed = int(z["n_steps"]), int(z["seed"])
    opt = {"lr": float(z["opt_lr"]), "lrd": float(z["opt_lrd"]), "betas": tuple(float(x) for x in z["opt_betas"])}
    eng = HipEngine(spec)
    run = SVIRunner(eng, opt, mode="parity", seed=seed)
    losses = np.array([run.step() for _ in range(n)])
    l64, l32 = z["loss64"], z["loss32"]
    rel_hip, rel_32 = np.abs(losses - l64) / np.abs(l64), np.abs(l32 - l64) / np.abs(l64)
    assert rel_hip[:5].max() <= 1e-5, rel_hip[:5]
    # A gene that crosses the relu kink of ElogU makes the loss of ONE step jump (1e-2 relative was observed, in the float32
    # oracle as in the HIP run, at different steps): the step-by-step yardstick of the short tests cannot be used over 1 500
    # steps.  Held instead: the worst step, the typical step and the converged level against the float32 oracle's own.
    assert rel_hip.max() <= max(1e-5, 4 * rel_32.max()), (rel_hip.max(), rel_32.max())
    assert np.median(rel_hip) <= max(1e-6, 4 * np.median(rel_32)), (np.median(rel_hip), np.median(rel_32))
    assert abs(losses[-100:].mean() - l64[-100:].mean()) <= max(1e-5, 4 * abs(l32[-100:].mean() - l64[-100:].mean()) / abs(l64[-100:].mean())) * abs(l64[-100:].mean())
    got = {k: v.detach().cpu().numpy() for k, v in eng.named().items()}
    H.assert_params_track_oracle(got, {k: z["par64_" + k] for k in got}, {k: z["par32_" + k] for k in got},
                                 report=f"{name}: final loss rel err {rel_hip[-1]:.2e} (float32 oracle {rel_32[-1]:.2e})")
    # ... and against the REFERENCE'S OWN fit() of the same problem, seed and optimiser (ref_fitlong_<name>.npz: the unmodified
    # PhaseFitModel.fit / VelocityFitModel.fit of /root/reference run on oracle/pyro_shim by `make_oracle_fits.py --reference`;
    # velocity_inference_model.py:111-187, phase_inference_model.py:162-201) -- north_star: "posterior means within 1e-3 rel of
    # reference".  The reference computes in float32 like the engine; the yardstick for two float32 trajectories is the distance
    # the reference itself keeps from the float64 trajectory.
    zr = np.load(os.path.join(H.GOLDEN, f"ref_fitlong_{name}.npz"))
    assert str(zr["digest"]) == str(z["digest"]) and int(zr["seed"]) == seed and int(zr["n_steps"]) == n
    lref = zr["ref_losses"]
    rel_ref, ref_64 = np.abs(losses - lref) / np.abs(lref), np.abs(lref - l64) / np.abs(l64)
    assert rel_ref[:5].max() <= 1e-5, rel_ref[:5]
    assert rel_ref.max() <= max(1e-5, 4 * ref_64.max()), (rel_ref.max(), ref_64.max())
    assert np.median(rel_ref) <= max(1e-6, 4 * np.median(ref_64)), (np.median(rel_ref), np.median(ref_64))
    H.assert_params_track_oracle(got, {k: zr["reffit_" + k] for k in got}, {k: z["par64_" + k] for k in got},
                                 report=f"{name} vs the reference's own fit(): final loss rel err {rel_ref[-1]:.2e} (reference vs float64 oracle {ref_64[-1]:.2e})")
    eng.close()


def _perf_fit(spec, n, seed):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    opt = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / n), "betas": (0.80, 0.99)}
    eng = HipEngine(spec)
    run = SVIRunner(eng, opt, mode="perf", seed=seed)
    assert run.adam_impl == "fused3"
    run.run_perf(n)
    losses = np.array(run.perf_losses())
    assert eng.status() == (True, -1, 0) and np.isfinite(losses).all()
    named = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.named().items()}
    eng.close()
    return losses, named


def _circ(named, spec):
    from velocycle_amd.utils import circular_corrcoef
    xy = named["ϕxy_locs"]
    return circular_corrcoef(np.arctan2(xy[:, 1], xy[:, 0]), spec.truth["phis"].cpu().numpy())


@pytest.mark.parametrize("name,omegas", [("vjoint_3000x200", (0.4,)), ("vjoint2_1500x200", (0.4, 0.3))])
def test_perf_mode_joint_fit_recovers_simulated_speed_and_phases(name, omegas):
    """Nothing conditioned, mean-field guide, Philox eps, fused three-launch step: after 1 500 steps the constant term of
    νω per condition agrees with the float64 oracle's own converged fit of the same problem (different eps stream) within
    3 posterior standard deviations (or 3 %), lies within 50 % of the simulated ω (see the module docstring), the ratio of the
    two samples' speeds within 10 % of the simulated 0.75, and the fitted phases follow the simulated ones."""
    z = np.load(os.path.join(H.GOLDEN, f"oracle_fit_{name}.npz"))
    spec = G.make_spec(name)
    losses, named = _perf_fit(spec, 1500, seed=5)
    w, sd = named["νω_locs"][:, 0], np.exp(named["νω_scales"][:, 0])
    w_orc = z["par64_νω_locs"].reshape(named["νω_locs"].shape)[:, 0]
    print(f"\n[{name}] omega fitted {w} +- {sd}; simulated {omegas}; float64 oracle fit {w_orc}; circ corr {_circ(named, spec):.4f}; "
          f"loss {losses[0]:.1f} -> {losses[-1]:.1f} (oracle {z['loss64'][-1]:.1f})")
    for x in range(len(omegas)):
        assert abs(w[x] - w_orc[x]) <= max(3 * sd[x], 0.03 * abs(w_orc[x])), (x, w[x], w_orc[x], sd[x])
        assert abs(w[x] - omegas[x]) <= 0.5 * omegas[x], (x, w[x], omegas[x])
    if len(omegas) == 2:
        assert abs(w[1] / w[0] - omegas[1] / omegas[0]) <= 0.1 * omegas[1] / omegas[0], (w, omegas)
    assert _circ(named, spec) > 0.95
    assert abs(losses[-100:].mean() - z["loss64"][-100:].mean()) <= 2e-3 * abs(z["loss64"][-100:].mean())


def test_perf_mode_tutorial_flow_recovers_two_sample_speeds():
    """The tutorials' two-stage flow through the drop-in API in its default performance mode: phase fit on the spliced counts
    -> velocity fit (default LRMN guide) conditioned on ϕxy, ν, Δν, shape_inv of the phase fit, two samples simulated with
    ω = 0.4 and 0.3; `speed_pyro.means` (the tutorials' headline output, Tutorial_Capolupo cell 63) returns their ratio
    (measured 0.753 against 0.75) and speeds of the simulated order of magnitude (the absolute scale is set by the γ prior,
    module docstring)."""
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.fit_models import PhaseFitModel, VelocityFitModel
    from velocycle_amd.optim import ClippedAdam
    from velocycle_amd.utils import circular_corrcoef
    from velocycle_amd.workloads import make_velocity_spec
    sp = G.make_spec("vjoint2_1500x200")                  # the stored two-sample data set (ω = 0.4 / 0.3)
    ad = AnnDataLite(sp.S.t().numpy(), sp.U.t().numpy())
    ad.obs["batch"] = [f"s{int(b)}" for b in sp.truth["batch"]]
    cyc = C.Cycle.from_array(sp.mu_nu.T.numpy(), sp.sd_nu.T.numpy(), list(ad.var.index))
    ph = C.Phases.from_array(sp.phixy_prior.T.numpy(), cell_names=list(ad.obs.index))
    Db = P.make_design_matrix(ad, ids="batch")
    n = 1500
    opt = lambda: ClippedAdam({"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)})
    torch.manual_seed(3)
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1)
    pf = PhaseFitModel(mp, num_samples=50, n_per_bin=50)
    pf.fit(opt(), num_steps=n, verbose=False)
    est = np.arctan2(pf.phis_pyro[1], pf.phis_pyro[0])
    cc = circular_corrcoef(est, sp.truth["phis"].cpu().numpy())
    cond = {"ϕxy": pf.phase_pyro.phi_xy_tensor.T, "ν": pf.cycle_pyro.means_tensor.T.unsqueeze(-2),
            "Δν": torch.tensor(pf.delta_nus), "shape_inv": torch.tensor(pf.disp_pyro).unsqueeze(-1)}
    spd = C.AngularSpeed.trivial_prior(condition_names=["s0", "s1"], harmonics=0)
    from velocycle_amd import pyro_compat as pyro
    pyro.clear_param_store()                              # as the tutorials do between the two stages
    mv = P.preprocess_for_velocity_estimation(ad, pf.cycle_pyro, pf.phase_pyro, spd, Db.float(), Db.float(), n_harmonics=1,
                                              count_factor=mp.count_factor, ω_n_harmonics=0, condition_on=cond)
    vf = VelocityFitModel(mv, condition_on=cond, num_samples=200, n_per_bin=50)
    vf.fit(opt(), num_steps=n, verbose=False)
    assert vf._runner.adam_impl == "fused3" and "vu_" in vf.engine.stats["main_kernel"]
    w = np.asarray(vf.speed_pyro.means, dtype=np.float64).reshape(-1)
    sd = np.asarray(vf.speed_pyro.stds, dtype=np.float64).reshape(-1)
    print(f"\n[tutorial flow, 2 x 1500 x 200] phase circ corr {cc:.4f}; omega {w} +- {sd}; simulated (0.4, 0.3); "
          f"ratio {w[1] / w[0]:.3f} (simulated 0.75)")
    assert cc > 0.95
    for x, truth in enumerate((0.4, 0.3)):
        assert 0.5 * truth <= w[x] <= 1.7 * truth, (x, w[x], sd[x], truth)
    assert abs(w[1] / w[0] - 0.75) < 0.06
