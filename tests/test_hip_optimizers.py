"""GPU: the optimiser contract of fit().  The reference passes whatever PyroOptim it is given to pyro.infer.SVI
(velocity_inference_model.py:76-84,111): pyro.optim.ClippedAdam in the package tutorials, pyro.optim.Adam (= torch.optim.Adam) in
tutorials/1D_Pancreas_Analysis.ipynb cell 26.  Both run in the HIP kernels with their own arithmetic (vc_set_optimizer /
vc_adam_update), weight_decay included; anything else is refused by name."""
import numpy as np
import pytest
import torch

from oracle import velocycle_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,wd", [("adam", 0.0), ("adam", 0.03), ("clipped_adam", 0.03), ("clipped_adam", 0.0)])
def test_adam_update_kernel_equals_torch(kind, wd):
    """vc_adam_update on flat buffers, 40 steps with a device step counter, against torch.optim.Adam itself (kind "adam") and
    against the oracle's restatement of pyro's clipped_adam.py (kind "clipped_adam", weight decay behind the clamp).  The second
    half of the buffer is a FROZEN tensor (a parameter without a path to the loss: its gradient is zero, PyroOptim never steps
    it): with weight decay on it must not move either."""
    from velocycle_amd.engine import HipEngine
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_phase_nb.npz")
    eng = HipEngine(H.spec_from_fixture(z))
    g = torch.Generator().manual_seed(2)
    n, half = 1000, 500
    p0 = torch.randn(n, generator=g, dtype=torch.float64)
    args = {"lr": 0.02, "betas": (0.85, 0.98), "eps": 1e-7, "weight_decay": wd}
    if kind == "clipped_adam":
        args.update(lrd=0.98, clip_norm=1.5)
        ref = orc.ClippedAdam(dict(args))
    else:
        ref = orc.Adam(dict(args))
        tp = torch.nn.Parameter(p0[:half].clone())
        topt = torch.optim.Adam([tp], **args)
    par = {"live": p0[:half].clone(), "frozen": p0[half:].clone()}
    dev = eng.device
    p = p0.float().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    frozen = torch.zeros(n, dtype=torch.uint8, device=dev)
    frozen[half:] = 1
    t_dev = torch.zeros(1, dtype=torch.int64, device=dev)
    for t in range(40):
        gr = torch.randn(half, generator=g, dtype=torch.float64) * (1 + 0.1 * t) + 0.2
        par = ref.step(par, {"live": gr, "frozen": torch.zeros(n - half, dtype=torch.float64).as_subclass(orc._NoPath)})
        if kind == "adam":
            tp.grad = gr.clone()
            topt.step()
        t_dev += 1
        gfull = torch.cat([gr, torch.zeros(n - half, dtype=torch.float64)]).float().to(dev)
        eng.adam_update(kind, p, gfull, m, v, args["lr"], args.get("lrd", 1.0), 0.85, 0.98, 1e-7, args.get("clip_norm", float("inf")), wd,
                        frozen=frozen, t=t + 1, t_dev=t_dev)
    torch.cuda.synchronize()
    want = torch.cat([par["live"], par["frozen"]])
    assert torch.allclose(p.double().cpu(), want, rtol=3e-5, atol=3e-6), (p.double().cpu() - want).abs().max()
    assert torch.equal(p[half:].cpu(), p0[half:].float())                    # the frozen half has not moved at all
    if kind == "adam":
        assert torch.allclose(par["live"], tp.data, rtol=1e-12, atol=1e-13)
    eng.close()


def _run(spec, args, impl, n, seed=9):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    e = HipEngine(spec)
    r = SVIRunner(e, dict(args), mode="perf", seed=seed, adam_impl=impl)
    flat0 = e.params.detach().clone()
    par0 = {k: v.detach().cpu().clone() for k, v in e.named().items()}
    r.run_perf(n)
    out = dict(p=e.params.clone().cpu(), l=np.array(r.perf_losses()), m=r.opt.m.clone().cpu(), named={k: v.detach().cpu().numpy().astype(np.float64) for k, v in e.named().items()},
               flat0=flat0, par0=par0, status=e.status(), kind=r.opt.kind)
    e.close()
    return out


@pytest.mark.parametrize("case", ["vel_mf_joint", "vel_lrmn_cond", "phase_nb_dnu2", "vel_mf_joint_dnu2"])
@pytest.mark.parametrize("args", [{"_kind": "adam", "lr": 0.02, "betas": (0.8, 0.99)},
                                  {"_kind": "adam", "lr": 0.02, "betas": (0.8, 0.99), "weight_decay": 0.01},
                                  {"_kind": "clipped_adam", "lr": 0.03, "lrd": 0.99, "betas": (0.8, 0.99), "weight_decay": 0.01}])
def test_every_step_structure_applies_the_chosen_optimiser(case, args):
    """Adam / weight decay through the fused two- or three-launch step, the unfused HIP sequence ("hip": vc_adam_update), the
    merged K_fin + optimiser launch ("fused") and the PyTorch-op optimiser ("torch"): one trajectory; and that trajectory is the
    float64 oracle's on the same Philox draws with torch's Adam / pyro's ClippedAdam restated (orc.Adam / orc.ClippedAdam)."""
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    n = 20
    runs = {impl: _run(spec, args, impl, n) for impl in ("fused3", "hip", "fused")}
    assert runs["hip"]["kind"] == args["_kind"]
    if spec.guide == "lrmn" and args.get("weight_decay"):
        # Pyro's own answer: `cov_factor` is initialised to clip(N(0, 0.02), min=0) (velocity_inference_guide.py:91-92), its zeros are
        # -inf in the unconstrained space PyroOptim steps in, and `grad.add(p, alpha=weight_decay)` makes them NaN -- the reference
        # cannot run weight decay on the LRMN guide, and neither does the engine pretend to: every structure latches the NaN
        for impl, r in runs.items():
            assert not r["status"][0] and r["status"][1] >= 0, (impl, r["status"])
        return
    for impl, r in runs.items():          # the HIP structures against each other: float32 rounding of re-associated sums
        assert r["status"] == (True, -1, 0) and len(r["l"]) == n
        base = runs["hip"]
        assert np.allclose(r["l"], base["l"], rtol=1e-6, atol=0), (impl, np.abs(r["l"] / base["l"] - 1).max())
        a, b = r["p"].double().numpy(), base["p"].double().numpy()
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin) and np.allclose(a[fin], b[fin], rtol=2e-4, atol=2e-5), (impl, np.abs(a[fin] - b[fin]).max())
    # the PyTorch-op optimiser (x / c where the kernels multiply by 1 / c, lerp_ instead of two products): the same update to
    # float32 rounding over three steps (Adam's m / sqrt(v) amplifies such rounding over a long run wherever a gradient is near zero)
    t3, h3 = _run(spec, args, "torch", 3), _run(spec, args, "hip", 3)
    assert np.allclose(t3["l"], h3["l"], rtol=1e-6, atol=0)
    a, b = t3["p"].double().numpy(), h3["p"].double().numpy()
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin) and np.allclose(a[fin], b[fin], rtol=1e-4, atol=1e-5), np.abs(a[fin] - b[fin]).max()
    got = runs["fused3"]
    eps = H.philox_eps_list(spec, got["flat0"], 9, n)
    oargs = {k: v for k, v in args.items() if k != "_kind"}
    mk = orc.Adam if args["_kind"] == "adam" else orc.ClippedAdam
    p64 = H.problem_from_spec(spec, torch.float64)
    l64, par64 = orc.fit(p64, oargs, n, eps_list=[{k: v.double() for k, v in e.items()} for e in eps],
                         params={k: v.double().clone() for k, v in got["par0"].items()}, opt=mk(dict(oargs)))
    l32, par32 = orc.fit(p64.to(torch.float32), oargs, n, eps_list=[{k: v.float() for k, v in e.items()} for e in eps],
                         params={k: v.float().clone() for k, v in got["par0"].items()}, opt=mk(dict(oargs)))
    l64, l32 = np.array(l64), np.array(l32)
    rel, rel32 = np.abs(got["l"] - l64) / np.abs(l64), np.abs(l32 - l64) / np.abs(l64)
    # The optimiser's arithmetic is pinned by the first five steps (1e-5: a missing weight decay or a misplaced eps shows there) and
    # by the kernel test above; behind them two float32 evaluations of a clamped, weight-decayed Adam flow drift apart -- measured
    # 4.3e-5 ... 1.1e-4 on two boxes where the float32 oracle itself is at 5e-6 ... 1e-4 of the float64 run
    assert rel[:5].max() <= 1e-5 and (rel <= np.maximum(2e-4, 8 * np.maximum.accumulate(rel32))).all(), (rel, rel32)
    H.assert_params_track_oracle(got["named"], {k: v.numpy() for k, v in par64.items()}, {k: v.double().numpy() for k, v in par32.items()})
    # ... and it is NOT what the other optimiser would have done (the contract used to run Adam objects as ClippedAdam)
    other = dict(args, _kind="clipped_adam" if args["_kind"] == "adam" else "adam")
    other.pop("lrd", None)
    o = _run(spec, other, "fused3", n)
    assert np.abs(o["p"][4:].double().numpy()[np.isfinite(o["p"][4:].numpy())] - got["p"][4:].double().numpy()[np.isfinite(got["p"][4:].numpy())]).max() > 1e-4


def test_fit_accepts_adam_objects_and_refuses_what_it_does_not_implement():
    from tests.test_fit_continue import _metaparams
    from velocycle_amd import pyro_compat as pyro
    z = H.load_fixture(f"{H.GOLDEN}/ref_fit_continue_vel_mf_joint.npz")
    mp, cond, Cls = _metaparams(z)
    fit = Cls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
    fit.fit(pyro.optim.Adam({"lr": 0.01, "betas": (0.8, 0.99)}), num_steps=12, verbose=False, seed=3)
    assert fit._runner.opt.kind == "adam" and fit._runner.opt.clip == float("inf") and fit._runner.opt.lrd == 1.0
    assert np.isfinite(fit.losses).all() and len(fit.losses) == 12

    class PyroOptimOfTorch:            # what a real pyro.optim.Adam looks like from outside
        def __init__(self, ctor, args):
            self.pt_optim_constructor, self.pt_optim_args = ctor, args
    pyro.clear_param_store()
    fit2 = Cls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
    fit2.fit(PyroOptimOfTorch(torch.optim.Adam, {"lr": 0.01, "betas": (0.8, 0.99)}), num_steps=12, verbose=False, seed=3)
    assert fit2._runner.opt.kind == "adam" and np.array_equal(fit2.losses, fit.losses)
    pyro.clear_param_store()
    for bad, exc in ((PyroOptimOfTorch(torch.optim.SGD, {"lr": 0.1}), TypeError), (PyroOptimOfTorch(torch.optim.RMSprop, {}), TypeError),
                     (pyro.optim.Adam({"amsgrad": True}), NotImplementedError), (pyro.optim.ClippedAdam({"lr": 0.1, "nesterov": True}), TypeError)):
        with pytest.raises(exc):
            Cls(mp, condition_on=cond, num_samples=4, n_per_bin=2).fit(bad, num_steps=2, verbose=False)
