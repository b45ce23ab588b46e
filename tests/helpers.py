"""Shared test helpers: fixtures -> oracle Problem / product ModelSpec."""
import glob
import os

import numpy as np
import torch

from oracle import velocycle_oracle as orc
from velocycle_amd.spec import ModelSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP_CASES = sorted(os.path.basename(p)[len("ref_step_"):-4] for p in glob.glob(os.path.join(GOLDEN, "ref_step_*.npz")))
FIT_CASES = sorted(os.path.basename(p)[len("ref_fit_"):-4] for p in glob.glob(os.path.join(GOLDEN, "ref_fit_*.npz"))
                   if not os.path.basename(p).startswith("ref_fit_continue_"))
CONTINUE_CASES = sorted(os.path.basename(p)[len("ref_fit_continue_"):-4] for p in glob.glob(os.path.join(GOLDEN, "ref_fit_continue_*.npz")))

_TENSOR_FIELDS = ["S", "U", "count_factor", "Db", "D", "mu_nu", "sd_nu", "phixy_prior", "mu_gamma", "sd_gamma",
                  "mu_beta", "sd_beta", "mu_nuw", "sd_nuw"]
_SCALAR_FIELDS = ["kind", "guide", "noisemodel", "with_delta_nu", "H", "Hw", "mu_dnu", "gamma_alpha", "gamma_beta",
                  "sigma_ln_s", "sigma_ln_u", "rho_mean", "rho_std", "rho_scale", "rho_rank"]


def load_fixture(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def _fields(z, dtype):
    kw = {}
    for f in _TENSOR_FIELDS:
        if "in_" + f in z:
            kw[f] = torch.tensor(z["in_" + f]).to(dtype)
    for f in _SCALAR_FIELDS:
        if "in_" + f in z:
            v = z["in_" + f].item()
            kw[f] = v
    if "in_sd_dnu" in z:
        v = z["in_sd_dnu"]
        kw["sd_dnu"] = float(v) if v.ndim == 0 else torch.tensor(v).to(dtype)
    kw["condition_on"] = {k[len("cond_"):]: torch.tensor(v).to(dtype) for k, v in z.items() if k.startswith("cond_")}
    kw["with_delta_nu"] = bool(kw["with_delta_nu"])
    return kw


def problem_from_fixture(z, dtype=torch.float64) -> orc.Problem:
    return orc.Problem(**_fields(z, dtype))


def spec_from_fixture(z) -> ModelSpec:
    return ModelSpec(**_fields(z, torch.float32))


def spec_from_problem(p: orc.Problem) -> ModelSpec:
    kw = {}
    for k, v in p.__dict__.items():
        if isinstance(v, torch.Tensor):
            kw[k] = v.float()
        elif k == "condition_on":
            kw[k] = {a: b.float() for a, b in v.items()}
        else:
            kw[k] = v
    return ModelSpec(**kw)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / scale)


def problem_from_spec(spec, dtype=torch.float64) -> orc.Problem:
    """The oracle's view of a product ModelSpec (CPU, `dtype`)."""
    kw = {}
    for k, v in spec.__dict__.items():
        if k in ("truth", "S_csr", "U_csr"):
            continue
        kw[k] = v.detach().cpu().to(dtype) if (isinstance(v, torch.Tensor) and v.is_floating_point()) else v
    kw["condition_on"] = {k: v.detach().cpu().to(dtype) for k, v in spec.condition_on.items()}
    return orc.Problem(**kw)


def assert_step_matches_oracle(eng, spec, eps, loss_rtol=1e-5, grad_rtol=2e-3):
    """One ELBO + gradient evaluation already launched on `eng` with the host eps dict `eps`: loss within `loss_rtol`
    of the float64 oracle, every gradient block within `grad_rtol` of its max-norm -- or no worse than 4x the error of
    the oracle's own float32 run (= what the reference computes in) where the 1/(z + 1e-5) relu kink amplifies rounding."""
    torch.cuda.synchronize()
    p64 = problem_from_spec(spec, torch.float64)
    par = {n: v.detach().cpu().double() for n, v in eng.named().items()}
    e64 = {k: v.double() for k, v in eps.items() if not k.startswith("_")}
    l64, g64, _, _ = orc.loss_and_grads(p64, par, e64)
    _, g32, _, _ = orc.loss_and_grads(p64.to(torch.float32), {k: v.float() for k, v in par.items()},
                                      {k: v.float() for k, v in e64.items()})
    assert abs(eng.loss() - l64) <= loss_rtol * abs(l64), (eng.loss(), l64)
    for name, got in eng.named(eng.grad).items():
        want = g64[name].numpy()
        fin = np.isfinite(want)
        err = np.abs(got.cpu().numpy()[fin] - want[fin]).max()
        ref32 = np.abs(g32[name].numpy().astype(np.float64)[fin] - want[fin]).max()
        strict = grad_rtol * max(np.abs(want[fin]).max(), 1e-3)
        assert err <= max(strict, 4 * ref32), (name, err, ref32)
        CLAUSE_STATS["blocks"] += 1
        if err > strict:
            # the block passed only because the float32 oracle itself is this far from float64 (the 1 / (z + 1e-5) relu kink):
            # counted and printed, so that a suite that leans on this clause says so (pytest -s / the summary line of conftest.py)
            CLAUSE_STATS["by_ref32_clause"].append((name, float(err / max(np.abs(want[fin]).max(), 1e-3)), float(ref32 / max(np.abs(want[fin]).max(), 1e-3))))
            print(f"[assert_step_matches_oracle] block {name!r} passed through the 4 x float32-oracle clause only: err {err:.3e} "
                  f"(strict bar {strict:.3e}), float32 oracle's own error {ref32:.3e}")
    return l64, g64


# how often assert_step_matches_oracle's second clause (<= 4 x the float32 oracle's own error) was what let a gradient block pass;
# tests/conftest.py prints the tally at the end of a run
CLAUSE_STATS = {"blocks": 0, "by_ref32_clause": []}


def assert_trajectory_within_float32_spread(spec, opt, n, seed, losses, named_params, snapshots=None):
    """SURVEY §8(d) ELBO-match over n SVI steps on the same host eps stream: the first steps agree with the float64
    oracle to 1e-5; afterwards float32 and float64 Adam trajectories separate by themselves, so the yardstick is the
    oracle's own float32 run (x4); fitted parameters within 1e-3 of each block's max-norm wherever float32 itself is.

    `snapshots` = {t: named parameters at the START of step t}, recorded by the caller: teacher forcing.  Adam's first steps
    move every parameter by ~lr * sign(gradient); where a gradient is zero to within float32 rounding (|g| = 23 next to a
    block max of 1e6 and a float32 error of 5e3 -- observed on this very workload) its sign is a coin toss in ANY float32
    evaluation, and the parameter starts 2 lr away from the float64 run's: a 1e-4 step in the loss one step later that says
    nothing about the kernels.  With snapshots the strict 1e-5 bar is therefore held where it tests the kernels -- the loss
    of step t against the float64 oracle evaluated AT the run's own parameters with the same draws -- and the free-running
    comparison allows 5e-4 (and the fitted parameters the quantile criterion of assert_params_track_oracle)."""
    p64 = problem_from_spec(spec, torch.float64)
    l64, par64 = orc.fit(p64, opt, n, seed=seed)
    l32, par32 = orc.fit(p64.to(torch.float32), opt, n, seed=seed)
    l64, l32, losses = np.array(l64), np.array(l32), np.array(losses)
    rel_hip, rel_32 = np.abs(losses - l64) / np.abs(l64), np.abs(l32 - l64) / np.abs(l64)
    assert rel_hip[0] <= 1e-5, rel_hip[:5]
    allow = 1e-5
    if snapshots:
        from velocycle_amd.rng import draw_eps
        g = torch.Generator().manual_seed(seed)
        draw_eps(spec, g)                                   # the guide's warm-up draw (SVIRunner parity mode, orc.fit)
        eps_t = [draw_eps(spec, g) for _ in range(max(snapshots) + 1)]
        for t, par in sorted(snapshots.items()):
            e64 = {k: v.double() for k, v in eps_t[t].items() if not k.startswith("_")}
            l_tf, _, _, _ = orc.loss_and_grads(p64, {k: v.detach().cpu().double() for k, v in par.items()}, e64)
            assert abs(losses[t] - l_tf) <= 1e-5 * abs(l_tf), (t, losses[t], l_tf)
        allow = 5e-4
    else:
        assert rel_hip[:5].max() <= 1e-5, rel_hip[:5]
    # free-running comparison: step by step against the float32 oracle's accumulated drift -- or, with teacher forcing (the
    # run may have left the float64 trajectory a few steps EARLIER than the float32 oracle happened to), against its largest
    # (x 8 there: two float32 runs of a flow that has left the float64 one are as far from each other as from it)
    yard = (8 * np.full_like(rel_32, rel_32.max())) if snapshots else 4 * np.maximum.accumulate(rel_32)
    assert (rel_hip <= np.maximum(allow, yard)).all(), (rel_hip.max(), rel_32.max())
    if snapshots:
        assert_params_track_oracle({k: v.detach().cpu().numpy() for k, v in named_params.items()},
                                   {k: v.numpy() for k, v in par64.items()}, {k: v.double().numpy() for k, v in par32.items()})
        return
    for k, v in named_params.items():
        want, got = par64[k].numpy(), v.detach().cpu().numpy().astype(np.float64)
        ref32 = par32[k].double().numpy()
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(got), fin), k
        if not fin.any():
            continue
        scale = max(np.abs(want[fin]).max(), 1e-2)
        err, spread = np.abs(got[fin] - want[fin]).max(), np.abs(ref32[fin] - want[fin]).max()
        assert err <= max(1e-3 * scale, 4 * spread), (k, err, spread, scale)


def philox_eps_list(spec, params_flat, seed, n):
    """The standard-normal draws the performance path uses at steps 0..n-1 (Philox4x32-10 keyed by (seed, step, index)),
    rebuilt INDEPENDENTLY of the run under test: a second engine's sampling kernel at (seed, t), read back through
    vc_read_site(eps).  Returned as the oracle's per-site eps dicts (float64, CPU)."""
    from velocycle_amd.engine import HipEngine
    eng = HipEngine(spec)
    eng.params.copy_(params_flat.to(eng.device))
    shapes = {"ν": (spec.Ng, spec.Nh), "νω": (spec.Nx, spec.Nhw), "ϕxy": (spec.Nc, 2)}
    out = []
    for t in range(n):
        eng.sample_guide(eps=None, seed=seed, step=t)
        flat = eng.read_site("eps")
        out.append({k: flat[o:o + s].double().reshape(shapes.get(k, (s,))) for k, (o, s) in eng.eps_slices.items()})
    eng.close()
    return out


def oracle_replay(spec, opt, par0_named, eps_list, dtype=torch.float64):
    """orc.fit on explicit initial parameters and eps draws: (losses, final unconstrained params)."""
    p = problem_from_spec(spec, dtype)
    par0 = {k: v.detach().cpu().to(dtype).clone() for k, v in par0_named.items()}
    eps = [{k: v.to(dtype) for k, v in e.items()} for e in eps_list]
    return orc.fit(p, opt, len(eps), eps_list=eps, params=par0)


def assert_params_track_oracle(got, par64, par32, frac=0.99, report=None):
    """Fitted parameters of a multi-step run against the float64 oracle trajectory on the same eps draws: per block,
    |got - want| <= max(1e-3 x the block's max-norm, 4 x the float32 oracle's own distance from float64).  ElogU has a relu
    kink with a 1/(z + 1e-5) factor behind it: a gene that crosses it sees its gradient change by orders of magnitude for a
    1e-4 change of its parameters (observed: -2137 vs +174 one step after a 1.7e-4 difference), so single elements of a
    float32 trajectory -- any float32 trajectory, the reference's own included -- can leave the float64 one by O(lr) per step.
    Blocks of >= 100 elements therefore have to hold the bar on `frac` of their elements (and may not do worse than the float32
    oracle by more than 1 - frac); small blocks on all of them.  Returns {block: (fraction within, max err / max-norm)}."""
    out = {}
    for k, g in got.items():
        want, ref32 = np.asarray(par64[k], dtype=np.float64), np.asarray(par32[k], dtype=np.float64)
        g = np.asarray(g, dtype=np.float64).reshape(want.shape)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(g), fin), k
        if not fin.any():
            continue
        scale = max(np.abs(want[fin]).max(), 1e-2)
        err, e32 = np.abs(g[fin] - want[fin]), np.abs(ref32[fin] - want[fin])
        tol = np.maximum(1e-3 * scale, 4 * e32.max())
        within, within32 = float((err <= tol).mean()), float((e32 <= 1e-3 * scale).mean())
        out[k] = (within, float(err.max() / scale), within32, float(e32.max() / scale))
        need = 1.0 if err.size < 100 else min(frac, within32 - (1 - frac))
        assert within >= need, (k, within, need, float(err.max()), float(e32.max()), scale)
    if report is not None:
        print(f"\n[{report}] per block: fraction within tolerance, max |err| / max-norm (HIP | float32 oracle): "
              + ", ".join(f"{k} {a:.4f} {b:.1e} | {c:.4f} {d:.1e}" for k, (a, b, c, d) in out.items()))
    return out
