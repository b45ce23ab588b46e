"""CPU: the oracle restatement against the committed golden vectors (outputs of the reference's own
code, see tests/golden/make_golden.py) and, when the reference tree is present, against a live run."""
import numpy as np
import pytest
import torch

from oracle import velocycle_oracle as orc
from tests import helpers as H


def test_basis_and_direction_golden():
    z = H.load_fixture(f"{H.GOLDEN}/basis.npz")
    phi = torch.tensor(z["phi"])
    for Hn in (0, 1, 2, 3):
        for der in (0, 1):
            got = orc.fourier_basis(phi, Hn, der).numpy()
            assert np.allclose(got, z[f"basis_H{Hn}_der{der}"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(orc.pack_direction(torch.tensor(z["xy"])).numpy(), z["pack_direction"])
    with pytest.raises(ValueError):
        orc.fourier_basis(phi, 1, der=2)            # utils.py:437


@pytest.mark.parametrize("case", H.STEP_CASES)
def test_oracle_step_matches_reference_fixture(case):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    p = H.problem_from_fixture(z, torch.float32)
    par = {k[4:]: torch.tensor(v) for k, v in z.items() if k.startswith("par_")}
    eps = {k[4:]: torch.tensor(v) for k, v in z.items() if k.startswith("eps_")}
    loss, grads, _, _ = orc.loss_and_grads(p, par, eps)
    assert abs(loss - float(z["ref_loss"])) <= 1e-5 * abs(float(z["ref_loss"])) + 1e-3
    for k, g in grads.items():
        want = z["refgrad_" + k]
        assert np.allclose(g.numpy(), want, rtol=2e-3, atol=2e-3 * max(1.0, np.abs(want).max())), k


@pytest.mark.parametrize("case", H.FIT_CASES)
def test_oracle_fit_matches_reference_fixture(case):
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_{case}.npz")
    p = H.problem_from_fixture(z, torch.float32)
    opt = {"lr": float(z["opt_lr"]), "lrd": float(z["opt_lrd"]), "betas": tuple(float(x) for x in z["opt_betas"])}
    losses, par = orc.fit(p, opt, int(z["num_steps"]), seed=int(z["seed"]))
    assert np.allclose(losses, z["ref_losses"], rtol=1e-4, atol=1e-2)
    for k, v in par.items():
        want = z["reffit_" + k]
        fin = np.isfinite(want)
        assert np.allclose(v.numpy()[fin], want[fin], rtol=2e-3, atol=2e-3), k


@pytest.mark.parametrize("case", H.CONTINUE_CASES)
@pytest.mark.parametrize("scen", ["same", "new"])
def test_oracle_continued_fit_matches_reference_fixture(case, scen):
    """Two fits without clearing the param store (make_golden.py --continue): the oracle told the same story -- stored
    parameters as the start, the SAME optimiser carrying its state on ("same") or a fresh one plus the fresh ELBO object's
    extra guide pass ("new") -- against the reference's own losses and fitted parameters of both fits."""
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_continue_{case}.npz")
    p = H.problem_from_fixture(z, torch.float32)
    opt = {"lr": float(z["opt_lr"]), "lrd": float(z["opt_lrd"]), "betas": tuple(float(x) for x in z["opt_betas"])}
    n, seed = int(z["num_steps"]), int(z["seed"])
    o = orc.ClippedAdam(opt)
    l1, par1 = orc.fit(p, opt, n, seed=seed, opt=o)
    assert np.allclose(l1, z[f"{scen}_ref_losses1"], rtol=1e-4, atol=1e-2)
    l2, par2 = orc.fit(p, opt, n, seed=seed + 1, params={k: v.clone() for k, v in par1.items()}, warmup_draw=(scen == "new"),
                       opt=(o if scen == "same" else None))
    assert np.allclose(l2, z[f"{scen}_ref_losses2"], rtol=1e-4, atol=1e-2)
    for k, v in par2.items():
        want = z[f"{scen}_reffit2_" + k]
        fin = np.isfinite(want)
        assert np.allclose(v.numpy()[fin], want[fin], rtol=2e-3, atol=2e-3), k


def test_oracle_rejects_unknown_noise_model():
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_phase_nb.npz")
    p = H.problem_from_fixture(z)
    p.noisemodel = "Gaussian"
    par = orc.init_params(p)
    with pytest.raises(ValueError):
        orc.elbo_loss(p, par, orc.draw_eps(p, torch.Generator().manual_seed(0)))


def test_closed_form_nb_identities():
    """The identities the HIP kernel relies on, in float64 against autograd:
    d NB/d eta = r (k - mu)/(r + mu);  sum_c (r+k)/(r+mu) = n + (sum_c a)/r;  lgamma terms via histogram."""
    torch.manual_seed(0)
    k = torch.poisson(torch.rand(200) * 20).double()
    eta = torch.randn(200, dtype=torch.float64, requires_grad=True)
    r = torch.tensor(3.7, dtype=torch.float64, requires_grad=True)
    lp = orc.gamma_poisson_log_prob(r, r / torch.exp(eta), k).sum()
    lp.backward()
    mu = torch.exp(eta.detach())
    a = r.detach() * (k - mu) / (r.detach() + mu)
    assert torch.allclose(eta.grad, a, rtol=1e-10, atol=1e-10)
    rr = r.detach()
    vals, cnt = torch.unique(k, return_counts=True)
    hd = (cnt * (torch.digamma(rr + vals) - torch.digamma(rr))).sum()
    dr = -(torch.log(rr + mu)).sum() - len(k) - a.sum() / rr + len(k) * (torch.log(rr) + 1) + hd
    assert abs(dr - r.grad) < 1e-8 * abs(r.grad)
    hl = (cnt * (torch.lgamma(rr + vals) - torch.lgamma(rr))).sum()
    ll = (k * eta.detach() - (rr + k) * torch.log(rr + mu)).sum() + len(k) * rr * torch.log(rr) + hl \
        - torch.lgamma(k + 1).sum()
    assert abs(ll - lp.detach()) < 1e-9 * abs(lp.detach())
