"""GPU: `fit(loss=Trace_ELBO(num_particles=K))` (VERDICT r2 "missing" #3).  The reference hands the user's ELBO object to
SVI (velocity_inference_model.py:79,111; phase_inference_model.py:128,162); with K particles pyro draws K guide samples per
step, one after the other, and averages loss and gradients before the optimiser step.  Fixtures: the reference's own fit()
run with num_particles = 3 (tests/golden/make_golden.py --particles -> ref_fitK3_*.npz)."""
import numpy as np
import pytest
import torch

from oracle import velocycle_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
CASES = ["phase_nb", "vel_mf_joint", "vel_lrmn_cond"]


def _opt(z):
    return {"lr": float(z["opt_lr"]), "lrd": float(z["opt_lrd"]), "betas": tuple(float(x) for x in z["opt_betas"])}


@pytest.mark.parametrize("case", CASES)
def test_parity_mode_with_three_particles_matches_the_reference_fit(case):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    z = H.load_fixture(f"{H.GOLDEN}/ref_fitK3_{case}.npz")
    spec = H.spec_from_fixture(z)
    K, n = int(z["num_particles"]), int(z["num_steps"])
    eng = HipEngine(spec)
    run = SVIRunner(eng, _opt(z), mode="parity", seed=int(z["seed"]), num_particles=K)
    losses = [run.step() for _ in range(n)]
    assert np.allclose(losses, z["ref_losses"], rtol=1e-4, atol=1e-2), np.abs(np.array(losses) - z["ref_losses"]).max()
    for k, v in eng.named().items():
        want, got = z["fit64_" + k], v.cpu().numpy()
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), k
        assert np.allclose(got[fin], want[fin], rtol=1e-3, atol=1e-3), (k, np.abs(got[fin] - want[fin]).max())
    eng.close()


@pytest.mark.parametrize("case", ["vel_mf_joint", "phase_nb"])
def test_perf_mode_particles_average_the_philox_streams(case):
    """perf mode, K = 3: step t averages the ELBO over the Philox streams (seed, 3 t + k); loss and gradient of the first step
    against the float64 oracle on exactly those three draws, and the trajectory against orc.fit(num_particles=3)."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    z = H.load_fixture(f"{H.GOLDEN}/ref_fitK3_{case}.npz")
    spec = H.spec_from_fixture(z)
    K, n, seed = 3, 6, 99
    eng = HipEngine(spec)
    run = SVIRunner(eng, _opt(z), mode="perf", seed=seed, num_particles=K)
    assert run.adam_impl == "hip"
    flat0 = eng.params.detach().clone()
    par0 = {k: v.detach().cpu().clone() for k, v in eng.named().items()}
    run.run_perf(n)
    losses = np.array(run.perf_losses())
    eps = H.philox_eps_list(spec, flat0, seed, n * K)
    p64 = H.problem_from_spec(spec, torch.float64)
    l64, par64 = orc.fit(p64, _opt(z), n, eps_list=eps, params={k: v.double() for k, v in par0.items()}, num_particles=K)
    assert np.allclose(losses, l64, rtol=2e-5), np.abs(losses / np.array(l64) - 1).max()
    for k, v in eng.named().items():
        want, got = par64[k].numpy(), v.cpu().numpy().astype(np.float64)
        fin = np.isfinite(want)
        assert np.allclose(got[fin], want[fin], rtol=2e-3, atol=2e-3), (k, np.abs(got[fin] - want[fin]).max())
    eng.close()


@pytest.mark.parametrize("case", ["vel_mf_joint", "phase_nb", "vel_lrmn_cond"])
def test_particles_from_one_c_call_equal_the_host_loop(case):
    """vc_svi_run_particles (every launch of an n-step, K-particle run enqueued from one C call; the particles' gradients averaged
    by a kernel) against the host loop of K x vc_elbo_grad + PyTorch averaging + vc_clipped_adam it replaces
    (Tuning(particles_host_loop=True)): the same kernels on the same Philox streams -- parameters, moments and losses bit for bit."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    z = H.load_fixture(f"{H.GOLDEN}/ref_fitK3_{case}.npz")
    spec = H.spec_from_fixture(z)
    out = []
    # the C call in its three launch layouts (csrc/vc_engine.hip: all particles' K_pre / K_post as one launch each and K_fin +
    # average + ClippedAdam as one = the default; per particle on one stream; per particle on streams of their own), then the host loop
    from velocycle_amd.tuning import Tuning
    for host, layout in ((False, "batched"), (False, "serial"), (False, "streams"), (True, "batched")):
        eng = HipEngine(spec, tuning=Tuning(particles_host_loop=host, particles_layout=layout))
        run = SVIRunner(eng, _opt(z), mode="perf", seed=5, num_particles=3)
        run.run_perf(4)
        run.run_perf(5)
        out.append((eng.params.clone().cpu(), run.opt.m.clone().cpu(), run.opt.v.clone().cpu(), run.perf_losses(),
                    int(run.step_dev.item()), run.opt.t, eng.status()))
        eng.close()
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    b = out[-1]
    for a in out[:-1]:
        assert torch.equal(nz(a[0]), nz(b[0])) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
        assert a[3] == b[3] and len(a[3]) == 9 and a[4] == b[4] == 9 and a[5] == b[5] == 9 and a[6] == b[6] == (True, -1, 0)


def test_particles_on_the_run_time_sized_kernel_set():
    """The C call on a generic engine (Tuning(force_generic=True) on a fast-set fixture): the per-particle launch layout on the generic
    kernels, again bit for bit the host loop -- and a change of K on the same engine (the particle workspaces grow)."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    z = H.load_fixture(f"{H.GOLDEN}/ref_fitK3_vel_mf_joint.npz")
    spec = H.spec_from_fixture(z)
    from velocycle_amd.tuning import Tuning
    out = []
    for host in (False, True):
        eng = HipEngine(spec, tuning=Tuning(force_generic=True, particles_host_loop=host))
        assert eng.stats["generic"]
        run = SVIRunner(eng, _opt(z), mode="perf", seed=5, num_particles=2)
        run.run_perf(3)
        run.K = 4
        run.run_perf(3)
        out.append((eng.params.clone().cpu(), run.opt.m.clone().cpu(), run.perf_losses(), eng.status()))
        eng.close()
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    assert torch.equal(nz(out[0][0]), nz(out[1][0])) and torch.equal(out[0][1], out[1][1])
    assert out[0][2] == out[1][2] and len(out[0][2]) == 6 and out[0][3] == out[1][3] == (True, -1, 0)


def test_fit_reads_num_particles_from_the_loss_object():
    """The drop-in API: PhaseFitModel.fit(optimizer, loss=Trace_ELBO(num_particles=3)) reproduces the reference's fit with
    the same object; an object asking for vectorised particles is refused by name."""
    from velocycle_amd import containers as C, preprocessing as P, pyro_compat as pyro
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.fit_models import PhaseFitModel
    z = H.load_fixture(f"{H.GOLDEN}/ref_fitK3_phase_nb.npz")
    ad = AnnDataLite(z["in_S"].T, z["in_S"].T * 0)
    cyc = C.Cycle.from_array(z["in_mu_nu"].T, z["in_sd_nu"].T, list(ad.var.index))
    ph = C.Phases.from_array(z["in_phixy_prior"].T, cell_names=list(ad.obs.index))
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, torch.tensor(z["in_Db"].T), n_harmonics=int(z["in_H"]), with_delta_nu=False)
    elbo = pyro.infer.Trace_ELBO(num_particles=3)
    assert elbo.fresh
    fit = PhaseFitModel(mp, num_samples=2, n_per_bin=2)
    fit.fit(pyro.optim.ClippedAdam(_opt(z)), loss=elbo, num_steps=int(z["num_steps"]), verbose=False, mode="parity",
            seed=int(z["seed"]))
    assert np.allclose(fit.losses, z["ref_losses"], rtol=1e-4, atol=1e-2) and fit._runner.K == 3 and not elbo.fresh

    class Vec:
        num_particles, vectorize_particles = 3, True
    # vectorize_particles=True: the same estimator (K draws averaged).  Parity mode (the reference's host RNG order) refuses it by
    # name; perf mode runs it as the batched K-particle step -- the same numbers as Trace_ELBO(num_particles=3) on the same seed
    with pytest.raises(NotImplementedError, match="vectorize_particles"):
        PhaseFitModel(mp, num_samples=2, n_per_bin=2).fit(pyro.optim.ClippedAdam(_opt(z)), loss=Vec(), num_steps=1, verbose=False,
                                                          mode="parity", seed=1)
    out = []
    for elbo_k in (Vec(), pyro.infer.Trace_ELBO(num_particles=3), pyro.infer.Trace_ELBO(num_particles=3, vectorize_particles=True)):
        pyro.clear_param_store()
        f = PhaseFitModel(mp, num_samples=2, n_per_bin=2)
        f.fit(pyro.optim.ClippedAdam(_opt(z)), loss=elbo_k, num_steps=6, verbose=False, mode="perf", seed=5)
        assert f._runner.K == 3 and len(f.losses) == 6 and np.isfinite(f.losses).all()
        out.append(np.array(f.losses))
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[1], out[2])
