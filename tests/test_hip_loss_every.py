"""GPU: the opt-in `loss_every = k` (vc_set_loss_every; SURVEY.md section 5 "or every k steps in perf mode").  Default k = 1 is the
reference's behaviour (velocity_inference_model.py:118-121 reads the loss of every step) and everything else in the suite; here:
the gradient-only instantiation of the U-only kernel steps like the full one (same gradients to float32 rounding: mu is formed
as 2^eta * z instead of 2^(eta + log2 z)), those of the S+U / S-only kernels give the full kernels' gradient bits, the loss
appears at every k-th step and is NaN in between, and configurations without such an instantiation refuse by name."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
OPT = {"lr": 0.03, "lrd": 0.9995, "betas": (0.8, 0.99)}


def _run(spec, k, steps, seed=7):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    eng = HipEngine(spec)
    run = SVIRunner(eng, OPT, mode="perf", seed=seed, loss_every=k)
    run.run_perf(steps // 2)
    run.run_perf(steps - steps // 2)
    out = (eng.params.clone().cpu(), run.perf_losses(), eng.status(), eng.stats["main_kernel"])
    eng.close()
    return out


@pytest.mark.parametrize("cells,genes,k", [(3000, 300, 4), (6000, 500, 10)])
def test_tutorial_flow_with_the_loss_every_kth_step(cells, genes, k):
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(cells, genes, "vcond", 1, 1, seed=3, device="cuda")
    steps = 3 * k + 2
    p1, l1, st1, name = _run(spec, 1, steps)
    pk, lk, stk, _ = _run(spec, k, steps)
    assert "vu_nb" in name and st1 == stk == (True, -1, 0)
    assert len(l1) == len(lk) == steps and all(math.isfinite(x) for x in l1)
    for i, (a, b) in enumerate(zip(l1, lk)):
        if i % k == 0:
            assert abs(a - b) <= 2e-6 * abs(a), (i, a, b)          # the same trajectory up to float32 rounding of mu
        else:
            assert math.isnan(b), (i, b)
    fin = torch.isfinite(p1)
    d = (p1[fin] - pk[fin]).abs()
    # parameters after 3k + 2 ClippedAdam steps: equal up to the rounding noise two float32 runs of this flow show (a sign
    # flip of a near-zero gradient moves a parameter by 2 lr in one step: a handful of elements)
    assert float(d.median()) <= 1e-5 and float((d <= 2e-3).float().mean()) >= 0.99, (float(d.median()), float(d.max()))


@pytest.mark.parametrize("mode", ["vjoint", "phase"])
def test_models_that_learn_shape_inv_keep_their_gradient_bits(mode):
    """S+U / S-only kernels: log2(r + mu) feeds d / d shape_inv and stays in the gradient-only instantiation; what goes is the
    loss-only log2(zp) and the two loss accumulations per element -- parameters bit for bit the default run's, losses at every
    k-th step equal, NaN in between."""
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    spec = make_phase_spec(4000, 300, seed=2, device="cuda") if mode == "phase" else make_velocity_spec(4000, 300, "vjoint", 1, 1, seed=2, device="cuda")
    p1, l1, st1, name = _run(spec, 1, 14)
    pk, lk, stk, _ = _run(spec, 4, 14)
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    assert torch.equal(nz(p1), nz(pk)) and st1 == stk == (True, -1, 0)
    for i, (a, b) in enumerate(zip(l1, lk)):
        assert (a == b) if i % 4 == 0 else math.isnan(b), (i, a, b)


def test_loss_every_is_refused_where_no_gradient_only_kernel_exists(monkeypatch):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_velocity_spec
    import dataclasses
    base = make_velocity_spec(2000, 200, "vjoint", 1, 1, seed=1, device="cuda")
    eng = HipEngine(dataclasses.replace(base, noisemodel="Poisson"))          # count noise without a dispersion: no such kernel
    with pytest.raises(NotImplementedError, match="gradient-only"):
        SVIRunner(eng, OPT, mode="perf", seed=1, loss_every=5)
    run = SVIRunner(eng, OPT, mode="perf", seed=1)          # the default is untouched
    run.run_perf(3)
    assert all(math.isfinite(x) for x in run.perf_losses())
    eng.close()
    spec = make_velocity_spec(2000, 200, "vcond", 1, 1, seed=1, device="cuda")
    eng = HipEngine(spec)
    run = SVIRunner(eng, OPT, mode="perf", seed=1, loss_every=3)
    with pytest.raises(ValueError, match="every step"):
        run.step_with_loss()
    run.run_perf(4)
    # the engine outlives its runners: the next runner on it forms every loss again
    run2 = SVIRunner(eng, OPT, mode="perf", seed=1, init=False)
    run2.run_perf(5)
    assert all(math.isfinite(x) for x in run2.perf_losses())
    eng.close()


def test_fit_takes_loss_every():
    """VelocityFitModel.fit(..., loss_every=k): `losses` holds the loss at every k-th step and NaN in between, no NaN warning;
    modes that read every loss refuse."""
    import warnings
    from tests import helpers as H
    from tests.test_fit_continue import _metaparams
    from tests.test_fit_api import _opt
    from velocycle_amd import pyro_compat as pyro
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_continue_vel_lrmn_cond.npz")       # the tutorial flow's velocity stage (conditioned)
    mp, cond, Cls = _metaparams(z)
    pyro.clear_param_store()
    vf = Cls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)
        vf.fit(_opt(z), num_steps=21, verbose=False, seed=3, loss_every=5)
    l = np.asarray(vf.losses)
    assert l.shape == (21,) and np.all(np.isfinite(l[::5])) and np.all(np.isnan(np.delete(l, np.arange(0, 21, 5))))
    # the same fit with every loss: the formed losses agree (same trajectory up to float32 rounding)
    pyro.clear_param_store()
    vf1 = Cls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
    vf1.fit(_opt(z), num_steps=21, verbose=False, seed=3)
    assert np.allclose(np.asarray(vf1.losses)[::5], l[::5], rtol=1e-5)
    pyro.clear_param_store()
    vf2 = Cls(mp, condition_on=cond, num_samples=4, n_per_bin=2, early_exit=True)
    with pytest.raises(ValueError, match="loss of every step"):
        vf2.fit(_opt(z), num_steps=5, verbose=False, seed=3, loss_every=5)
