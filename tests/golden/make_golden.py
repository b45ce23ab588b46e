#!/usr/bin/env python
"""Generate the golden fixtures of tests/golden/ by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference; the GPU box does not have it).  The
reference package is imported unmodified from /root/reference/build/lib on top of `oracle/pyro_shim`
(pyro-ppl is not installable here).  What is written is DATA only: inputs, seeds, the
eps stream, and the reference's outputs (losses, gradients, fitted parameters, posterior summaries).

  basis.npz                 utils.torch_fourier_basis / pack_direction / unpack_direction outputs
  ref_step_<case>.npz       one Trace_ELBO.loss_and_grads of model_fn/guide_fn (loss + every gradient)
  ref_fit_<case>.npz        N steps of {Phase,Velocity}FitModel.fit (losses, final params, attributes)

While generating, every case is also evaluated with the oracle restatement (oracle/velocycle_oracle.py)
and the script ABORTS if the two disagree -- this is what pins the oracle.

Usage:  python tests/golden/make_golden.py [--real-pyro | --shim] [--check] [--continue | --particles | <case> ...]

  --real-pyro   run the reference on the installed pyro-ppl 1.8.x instead of oracle/pyro_shim (error if it is not importable);
                without a flag the real library is used whenever it is there (oracle/ref_loader.py)
  --check       write NOTHING: regenerate every fixture in memory and diff it against the committed .npz (the tolerances of
                check()); exit status 1 on any disagreement.  `--check --real-pyro` is the one command that pins the Pyro
                boundary (Trace_ELBO, ClippedAdam, plate, poutine.condition / block) for whoever has the library; on the shim
                it proves that the committed fixtures are what this script produces.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import ref_loader                          # noqa: E402
ARGS = [a for a in sys.argv[1:] if a not in ("--real-pyro", "--shim", "--check")] if __name__ == "__main__" else []
CHECK = __name__ == "__main__" and "--check" in sys.argv[1:]
_backend = "auto"
if __name__ == "__main__" and "--real-pyro" in sys.argv[1:]:
    _backend = "real"
if __name__ == "__main__" and "--shim" in sys.argv[1:]:
    _backend = "shim"
vc = ref_loader.load_reference(_backend)
import pyro                                           # noqa: E402  (the real library or the shim: ref_loader.BACKEND)
from oracle import velocycle_oracle as orc            # noqa: E402
from velocycle_amd.simulate import simulate_counts    # noqa: E402
from velocycle_amd.anndata_lite import AnnDataLite    # noqa: E402

CANON = {  # reference param name -> canonical shape builder
    "ν_locs": lambda p: (p.Ng, p.Nh), "ν_scales": lambda p: (p.Ng, p.Nh),
    "Δν_locs": lambda p: (p.Nb, p.Ng), "ϕxy_locs": lambda p: (p.Nc, 2),
    "logγg_locs": lambda p: (p.Ng,), "logγg_scales": lambda p: (p.Ng,),
    "logβg_locs": lambda p: (p.Ng,), "logβg_scales": lambda p: (p.Ng,),
    "νω_locs": lambda p: (p.Nx, p.Nhw), "νω_scales": lambda p: (p.Nx, p.Nhw),
    "shape_inv_locs": lambda p: (p.Ng,), "loc": lambda p: (p.Ng + p.Nx * p.Nhw,),
    "cov_factor": lambda p: (p.Ng + p.Nx * p.Nhw, p.rho_rank), "cov_diag": lambda p: (p.Ng + p.Nx * p.Nhw,),
    "rho_real_loc": lambda p: (p.Ng,),
}


def build_inputs(Nc, Ng, H, n_batches, seed):
    """Synthetic data + priors built with the reference's own containers (tutorial cells 16-21)."""
    d = simulate_counts(Nc, Ng, omegas=(0.4, 0.3)[:n_batches], seed=seed)
    ad = AnnDataLite(d["S"].numpy(), d["U"].numpy())
    ad.obs["batch"] = [f"b{int(b)}" for b in d["batch"]]
    genes = list(ad.var.index)
    S = d["S"].numpy()
    cyc = vc.cycle.Cycle.trivial_prior(gene_names=genes, harmonics=H)
    nu0 = np.log(S.mean(0) + 0.05)
    nu0std = np.std(np.log(S + 1), axis=0) / 2 + 0.05
    rs = np.random.RandomState(seed)
    means = np.vstack([nu0] + [0.1 * rs.randn(len(genes)) for _ in range(2 * H)])
    stds = np.vstack([nu0std] + [0.5 * nu0std for _ in range(2 * H)])
    cyc.set_means(means)
    cyc.set_stds(stds)
    phi0 = d["phis"].numpy() + 0.3 * rs.randn(ad.n_obs)
    ph = vc.phases.Phases.from_array(np.stack([2.0 * np.cos(phi0), 2.0 * np.sin(phi0)]),
                                     cell_names=list(ad.obs.index))
    Db = vc.preprocessing.make_design_matrix(ad, ids="batch")
    return d, ad, cyc, ph, Db


MISMATCH = []          # --check: what differed from the committed fixtures


def save(fname, out):
    """Writes the fixture -- or, with --check, holds the freshly generated arrays against the committed file: float arrays within
    the tolerances check() uses for oracle == reference (2e-4 relative / absolute, the absolute part scaled by the block's
    largest element), everything else exactly."""
    path = os.path.join(OUT, fname)
    if not CHECK:
        np.savez_compressed(path, **out)
        return
    if not os.path.exists(path):
        MISMATCH.append(f"{fname}: no committed fixture")
        return
    z = np.load(path, allow_pickle=False)
    new = {k: np.asarray(v) for k, v in out.items()}
    for k in sorted(set(z.files) | set(new)):
        if k not in z.files or k not in new:
            MISMATCH.append(f"{fname}: key {k} only in the {'regenerated' if k in new else 'committed'} fixture")
            continue
        a, b = new[k], z[k]
        if a.shape != b.shape:
            MISMATCH.append(f"{fname}: {k} shape {a.shape} != committed {b.shape}")
        elif a.dtype.kind in "fc" and b.dtype.kind in "fc":
            a64, b64 = a.astype(np.float64), b.astype(np.float64)
            fin = np.isfinite(a64) & np.isfinite(b64)
            scale = np.abs(b64[fin]).max() if fin.any() else 0.0
            if not np.array_equal(np.isfinite(a64), np.isfinite(b64)) or \
                    not np.allclose(a64[fin], b64[fin], rtol=2e-4, atol=2e-4 * max(1.0, scale)):
                err = np.abs(a64[fin] - b64[fin]).max() if fin.any() else float("nan")
                MISMATCH.append(f"{fname}: {k} max abs difference {err:.3e} (block max {scale:.3e})")
        elif not np.array_equal(a, b):
            MISMATCH.append(f"{fname}: {k} differs (exact comparison)")
    print(f"[check] {fname}: {len(new)} arrays compared")


def ref_params_canonical(p):
    store = pyro.get_param_store()
    uncon = ref_loader.unconstrained_params(store)
    vals, grads = {}, {}
    for name in store.keys():
        u = uncon[name]
        shp = CANON[name](p)
        vals[name] = u.detach().reshape(shp).clone()
        grads[name] = (torch.zeros_like(u) if u.grad is None else u.grad).detach().reshape(shp).clone()
    return vals, grads


def problem_arrays(p, prefix="in_"):
    out = {}
    for k, v in p.__dict__.items():
        if isinstance(v, torch.Tensor):
            out[prefix + k] = v.numpy()
        elif isinstance(v, (int, float, bool, str)):
            out[prefix + k] = np.array(v)
    for k, v in p.condition_on.items():
        out["cond_" + k] = v.numpy()
    return out


def check(a, b, what, rtol=2e-4, atol=2e-4, block_rtol=0.0):
    """block_rtol > 0 (medium-size cases): the absolute tolerance is at least block_rtol x the block's largest element -- two
    float32 evaluations of a sum over hundreds of cells differ by rounding that scales with the sum's terms, not with each
    (possibly cancelling) result."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    if block_rtol > 0 and fin.any():
        atol = max(atol, block_rtol * np.abs(b[fin]).max())
    if not np.array_equal(np.isfinite(a), np.isfinite(b)) or not np.allclose(a[fin], b[fin], rtol=rtol, atol=atol):
        err = np.abs(a[fin] - b[fin]).max() if fin.any() else float("nan")
        raise SystemExit(f"ORACLE != REFERENCE for {what}: max abs err {err} (block max {np.abs(b[fin]).max() if fin.any() else float('nan')})")


# ------------------------------------------------------------------------------------------------
CASES = {
    # name: dict(kind, Nc, Ng, H, Hw, nb, noise, model_type, with_delta_nu, cond (list of sites), sdnu_tensor)
    "phase_nb":        dict(kind="phase", Nc=37, Ng=11, H=1, nb=1, noise="NegativeBinomial", wdn=False),
    "phase_nb_dnu2":   dict(kind="phase", Nc=23, Ng=9, H=2, nb=2, noise="NegativeBinomial", wdn=True, sdnu_tensor=True),
    "phase_poisson":   dict(kind="phase", Nc=29, Ng=7, H=1, nb=1, noise="Poisson", wdn=True),
    "phase_lognormal": dict(kind="phase", Nc=29, Ng=7, H=1, nb=1, noise="Lognormal", wdn=False),
    "vel_mf_joint":    dict(kind="velocity", Nc=37, Ng=11, H=1, Hw=1, nb=1, noise="NegativeBinomial",
                            model_type="normal", wdn=False),
    "vel_mf_joint_dnu2": dict(kind="velocity", Nc=23, Ng=9, H=2, Hw=1, nb=2, noise="NegativeBinomial",
                              model_type="normal", wdn=True),
    "vel_mf_cond":     dict(kind="velocity", Nc=37, Ng=11, H=1, Hw=0, nb=1, noise="NegativeBinomial",
                            model_type="normal", wdn=False, cond=["ϕxy", "ν", "shape_inv"]),
    "vel_lrmn_cond":   dict(kind="velocity", Nc=37, Ng=11, H=1, Hw=1, nb=1, noise="NegativeBinomial",
                            model_type="lrmn", wdn=False, cond=["ϕxy", "ν", "shape_inv"]),
    "vel_lrmn_cond_dnu2": dict(kind="velocity", Nc=23, Ng=9, H=1, Hw=0, nb=2, noise="NegativeBinomial",
                               model_type="lrmn", wdn=True, cond=["ϕxy", "ν", "Δν", "shape_inv"]),
    "vel_lrmn_joint":  dict(kind="velocity", Nc=29, Ng=7, H=1, Hw=1, nb=1, noise="NegativeBinomial",
                            model_type="lrmn", wdn=False),
    "vel_mf_poisson":  dict(kind="velocity", Nc=29, Ng=7, H=1, Hw=1, nb=1, noise="Poisson",
                            model_type="normal", wdn=False),
    "vel_mf_lognormal": dict(kind="velocity", Nc=29, Ng=7, H=1, Hw=1, nb=1, noise="Lognormal",
                             model_type="normal", wdn=False),
    # a MEDIUM single step: two gene blocks of the 4-genes-per-lane layout, several ragged cell tiles,
    # two harmonics, two batches with learned offsets, the default LRMN guide -- the toy cases above fit one block and one tile
    "vel_lrmn_joint_dnu2_med": dict(kind="velocity", Nc=700, Ng=300, H=2, Hw=1, nb=2, noise="NegativeBinomial",
                                    model_type="lrmn", wdn=True),
}
FIT_CASES = {"phase_nb": 25, "vel_mf_joint": 25, "vel_lrmn_cond": 25, "vel_mf_cond": 15,
             # the medium case through the reference's own fit(): two batches with learned offsets over 12 steps (round 5: the path
             # on which the engine folds the one-hot batch design per workgroup)
             "vel_lrmn_joint_dnu2_med": 12}


def make_case(name, c, seed=11):
    d, ad, cyc, ph, Db = build_inputs(c["Nc"], c["Ng"], c["H"], c["nb"], seed)
    Nc = ad.n_obs
    rs = np.random.RandomState(seed + 1)
    cond = {}
    if c["kind"] == "phase":
        kw = {}
        if c.get("sdnu_tensor"):
            s = torch.ones((c["nb"], c["Ng"], 1))
            s[0] = 0.001
            s[1:] = 0.1
            kw["σΔν"] = s
        mp = vc.preprocessing.preprocess_for_phase_estimation(
            ad, cyc, ph, Db, n_harmonics=c["H"], noisemodel=c["noise"], with_delta_nu=c["wdn"], **kw)
        FitCls = vc.phase_inference_model.PhaseFitModel
    else:
        spd = vc.angularspeed.AngularSpeed.trivial_prior(
            condition_names=[f"b{i}" for i in range(c["nb"])], harmonics=c["Hw"])
        if c["Hw"] == 1:
            spd.stds.loc["nu1_cos"] = [0.05] * c["nb"]
            spd.stds.loc["nu1_sin"] = [0.05] * c["nb"]
        cf = torch.tensor(np.log(ad.layers["spliced"].sum(1) / ad.layers["spliced"].sum(1).mean())).float()[None, None, :]
        for site in c.get("cond", []):
            if site == "ϕxy":
                cond[site] = ph.phi_xy_tensor.T + torch.tensor(0.05 * rs.randn(Nc, 2)).float()
            elif site == "ν":
                cond[site] = cyc.means_tensor.T.unsqueeze(-2) + torch.tensor(0.05 * rs.randn(c["Ng"], 1, 2 * c["H"] + 1)).float()
            elif site == "Δν":
                cond[site] = torch.tensor(0.01 * rs.randn(c["nb"], 1, 1, c["Ng"], 1)).float()
            elif site == "shape_inv":
                cond[site] = torch.tensor(rs.uniform(0.2, 1.0, (c["Ng"], 1))).float()
        mp = vc.preprocessing.preprocess_for_velocity_estimation(
            ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=c["H"], ω_n_harmonics=c["Hw"],
            count_factor=cf, noisemodel=c["noise"], with_delta_nu=c["wdn"], condition_on=cond,
            model_type=c.get("model_type", "lrmn"))
        FitCls = vc.velocity_inference_model.VelocityFitModel

    p64 = orc.problem_from_metaparams(mp, c["kind"], cond, dtype=torch.float64)
    p32 = p64.to(torch.float32)

    # ---- one step through the reference's model_fn / guide_fn --------------------------------
    fitm = FitCls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
    pyro.clear_param_store()
    torch.manual_seed(seed)
    elbo = pyro.infer.Trace_ELBO(num_particles=1)
    ref_loss, _ = elbo.loss_and_grads(fitm.model, fitm.guide, mp)
    ref_par, ref_grad = ref_params_canonical(p32)

    gen = torch.Generator().manual_seed(seed)
    warm = orc.draw_eps(p32, gen)           # the _guess_max_plate_nesting pass
    eps = orc.draw_eps(p32, gen)
    par32 = orc.init_params(p32, warm.get("_cov_factor_draw"))
    for k in ref_par:                        # initial parameter values agree
        check(par32[k], ref_par[k], f"{name}: init {k}", 1e-6, 1e-6)
    o_loss, o_grad, o_val, o_det = orc.loss_and_grads(p32, par32, eps)
    check(o_loss, ref_loss, f"{name}: loss", 1e-5, 1e-3)
    for k in ref_grad:
        check(o_grad[k], ref_grad[k], f"{name}: grad {k}", 2e-3, 2e-3, block_rtol=(2e-5 if c["Nc"] >= 200 else 0.0))
    # float64 oracle values = what the HIP path is compared with
    par64 = {k: v.double() for k, v in par32.items()}
    eps64 = {k: v.double() for k, v in eps.items()}
    l64, g64, v64, d64 = orc.loss_and_grads(p64, par64, eps64)
    out = problem_arrays(p32)
    out.update({"par_" + k: v.numpy() for k, v in par32.items()})
    out.update({"eps_" + k: v.numpy() for k, v in eps.items() if not k.startswith("_")})
    out.update({"refgrad_" + k: v.numpy() for k, v in ref_grad.items()})
    out.update({"grad64_" + k: v.numpy() for k, v in g64.items()})
    out.update({"val64_" + k: v.numpy() for k, v in v64.items()})
    out["ref_loss"] = np.array(ref_loss)
    out["loss64"] = np.array(l64)
    out["seed"] = np.array(seed)
    save(f"ref_step_{name}.npz", out)
    print(f"[step] {name}: ref loss {ref_loss:.4f}  oracle32 {o_loss:.4f}  oracle64 {l64:.4f}")

    # ---- N steps of the reference's own fit() ---------------------------------------------------
    if name in FIT_CASES:
        n = FIT_CASES[name]
        opt_args = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)}
        fitm = FitCls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
        pyro.clear_param_store()
        torch.manual_seed(seed)
        fitm.fit(pyro.optim.ClippedAdam(dict(opt_args)), loss=pyro.infer.Trace_ELBO(num_particles=1),
                 num_steps=n, verbose=False)
        ref_par, _ = ref_params_canonical(p32)
        o_losses, o_par = orc.fit(p32, opt_args, n, seed=seed)
        l64s, par64f = orc.fit(p64, opt_args, n, seed=seed)
        if c["Nc"] < 200:
            check(o_losses, fitm.losses, f"{name}: fit losses", 1e-4, 1e-2)
            for k in ref_par:
                check(o_par[k], ref_par[k], f"{name}: fitted {k}", 2e-3, 2e-3)
        else:
            # medium size: two float32 runs of the flow (the reference on the shim, the oracle) separate step by step where genes sit
            # on the relu kink of ElogU -- the yardstick is the distance the float32 oracle itself keeps from the float64 one
            lr_, lo_, l6_ = np.array(fitm.losses), np.array(o_losses), np.array(l64s)
            rel, spread = np.abs(lo_ / lr_ - 1), np.abs(lo_ / l6_ - 1)
            print(f"[fit ] {name}: oracle32 vs reference per step {np.array2string(rel, precision=1)}; oracle32 vs oracle64 {np.array2string(spread, precision=1)}")
            if rel[:2].max() > 1e-5 or (rel > np.maximum(1e-5, 4 * np.maximum.accumulate(np.maximum(spread, np.abs(lr_ / l6_ - 1))))).any():
                raise SystemExit(f"ORACLE != REFERENCE for {name}: fit losses beyond the float32 spread")
            for k in ref_par:
                a, b, c64 = o_par[k].numpy().astype(np.float64), ref_par[k].numpy().astype(np.float64), par64f[k].numpy()
                fin = np.isfinite(b)
                tol = max(1e-3 * np.abs(c64[fin]).max(), 4 * max(np.abs(a[fin] - c64[fin]).max(), np.abs(b[fin] - c64[fin]).max()))
                if not np.array_equal(np.isfinite(a), fin) or np.abs(a[fin] - b[fin]).max() > tol:
                    raise SystemExit(f"ORACLE != REFERENCE for {name}: fitted {k}: {np.abs(a[fin] - b[fin]).max()} > {tol}")
        fo = problem_arrays(p32)
        fo.update({"reffit_" + k: v.numpy() for k, v in ref_par.items()})
        fo.update({"fit64_" + k: v.numpy() for k, v in par64f.items()})
        fo["ref_losses"] = np.array(fitm.losses)
        fo["losses64"] = np.array(l64s)
        fo["num_steps"] = np.array(n)
        fo["seed"] = np.array(seed)
        for k, v in opt_args.items():
            fo["opt_" + k] = np.array(v)
        for attr in ("phis_pyro", "fourier_coef", "fourier_coef_sd", "disp_pyro", "delta_nus",
                     "log_gammas", "log_betas", "velocity_coef", "velocity_coef_sd"):
            if hasattr(fitm, attr):
                fo["attr_" + attr] = np.asarray(getattr(fitm, attr))
        if c["Nc"] >= 200:      # medium case: the trajectory only (the posterior arrays would be megabytes) -> ref_fitmed_<case>.npz
            save(f"ref_fitmed_{name}.npz", fo)
            print(f"[fit ] {name}: {n} steps, ref final loss {fitm.losses[-1]:.4f}, oracle {o_losses[-1]:.4f}")
            return
        # the reference's own posterior summaries (velocity_inference_model.py:236-262, phase_inference_model.py:248-265)
        # together with the draw-dependent inputs they were computed from, so that the engine's vc_expected_logs can be
        # checked against the reference's numbers without sharing its RNG stream
        post = fitm.posterior
        for k in ("ElogS", "ElogS2", "ElogU", "ElogU2"):
            if k in post:
                fo["post_" + k] = post[k].detach().numpy()
        fo["post_phis"] = fitm.phase_pyro.phis.detach().numpy()
        fo["post_cf_avg"] = np.array(float(fitm.metaparams_avg.count_factor.reshape(-1)[0]))
        if c["kind"] == "velocity":
            fo["post_gamma_mean"] = post["γg"].mean(0).reshape(-1).numpy()
            fo["post_logbeta_mean"] = post["logβg"].mean(0).reshape(-1).numpy()
            fo["post_nuw_mean"] = post["νω"].mean(0).reshape(p32.Nx, p32.Nhw).numpy()
            fo["post_omega_draws"] = post["ω"].reshape(post["ω"].shape[0], -1).numpy()
            fo["post_nuw_draws"] = post["νω"].reshape(post["νω"].shape[0], p32.Nx, p32.Nhw).numpy()
            fo["post_phi_draws"] = post["ϕ"].reshape(post["ϕ"].shape[0], -1).numpy()
        save(f"ref_fit_{name}.npz", fo)
        print(f"[fit ] {name}: {n} steps, ref final loss {fitm.losses[-1]:.4f}, oracle {o_losses[-1]:.4f}")


def make_basis():
    U = vc.utils
    phi = torch.tensor(np.linspace(-3.5, 7.0, 41), dtype=torch.float32)
    out = {"phi": phi.numpy()}
    for H in (0, 1, 2, 3):
        for der in (0, 1):
            ref = U.torch_fourier_basis(phi, num_harmonics=H, der=der)
            mine = orc.fourier_basis(phi, H, der)
            check(mine, ref, f"basis H={H} der={der}", 1e-6, 1e-6)
            out[f"basis_H{H}_der{der}"] = ref.numpy()
    xy = torch.tensor(np.random.RandomState(0).randn(50, 2), dtype=torch.float32)
    ref = U.pack_direction(xy)
    check(orc.pack_direction(xy), ref, "pack_direction", 0, 0)
    out["xy"] = xy.numpy()
    out["pack_direction"] = ref.numpy()
    out["unpack_direction"] = U.unpack_direction(ref, 1.0).numpy()
    save("basis.npz", out)
    print("[basis] ok")


def make_preprocess():
    """The MetaparContainer tensors the reference's preprocess_for_* build from a duck-typed AnnData."""
    d, ad, cyc, ph, Db = build_inputs(21, 6, 1, 2, seed=3)
    out = {"S": ad.layers["spliced"], "U": ad.layers["unspliced"], "batch": np.array(ad.obs["batch"]).astype(str),
           "cyc_means": cyc.means.values, "cyc_stds": cyc.stds.values, "phi_xy": ph.phi_xy.values}
    mp = vc.preprocessing.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1)
    for k in ("Db", "μνg", "σνg", "φxy_prior", "count_factor", "S", "U", "logS", "σΔν", "μΔν"):
        out["phase_" + k] = getattr(mp, k).numpy()
    spd = vc.angularspeed.AngularSpeed.trivial_prior(condition_names=["b0", "b1"], harmonics=1)
    mv = vc.preprocessing.preprocess_for_velocity_estimation(ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=1,
                                                             count_factor=mp.count_factor, ω_n_harmonics=1)
    for k in ("D", "Db", "ν", "μγ", "σγ", "μβ", "σβ", "μνω", "σνω", "μνg", "σνg", "φxy_prior", "count_factor", "S", "U",
              "logU", "σsgc"):
        out["vel_" + k] = getattr(mv, k).numpy()
    out["vel_model_type"] = np.array(mv.model_type)
    out["design"] = Db.numpy()
    # dense layers with NON-integer values (e.g. pre-normalised data): the reference's `.A` branch fails for an ndarray, its
    # `except` branch keeps the floats in the phase container (preprocessing.py:141-147) while the velocity container casts
    # to int64 in both branches (:243-252)
    ad2 = AnnDataLite(ad.layers["spliced"] * 0.5 + 0.25, ad.layers["unspliced"] * 0.75)
    ad2.obs["batch"] = list(ad.obs["batch"])
    mp2 = vc.preprocessing.preprocess_for_phase_estimation(ad2, cyc, ph, Db, n_harmonics=1)
    mv2 = vc.preprocessing.preprocess_for_velocity_estimation(ad2, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=1,
                                                              count_factor=mp2.count_factor, ω_n_harmonics=1)
    for k in ("S", "U", "logS", "count_factor"):
        out["nonint_phase_" + k] = getattr(mp2, k).numpy()
    for k in ("S", "U", "logU"):
        out["nonint_vel_" + k] = getattr(mv2, k).numpy()
    save("ref_preprocess.npz", out)
    print("[preprocess] ok")


def make_phase_prior():
    """Phases.from_pca_heuristic / max_corr / rotate of the reference (the step right before phase inference)."""
    d, ad, cyc, ph, Db = build_inputs(80, 12, 1, 1, seed=5)
    vc.preprocessing.normalize_total(ad) if hasattr(ad.layers["spliced"], "toarray") else None
    S = ad.layers["spliced"].astype(float)
    ad.layers["S_sz"] = (S.sum(1).mean() / np.maximum(S.sum(1), 1) * S.T).T
    out = {"S_sz": ad.layers["S_sz"], "umis": S.sum(1)}
    for tag, kw in (("a", dict(concentration=5.0, small_count=1)),
                    ("b", dict(concentration=1.0, small_count=0.1, zero_at_min_density=True, normalize_pcs=False))):
        p = vc.phases.Phases.from_pca_heuristic(ad, layer="S_sz", **kw)
        out["phixy_" + tag] = p.phi_xy.values
        shift, c, corr = p.max_corr(S.sum(1), npoints=50)
        out["maxcorr_" + tag] = np.array([shift, c])
        out["corr_" + tag] = np.array(corr)
        p.rotate(angle=-shift)
        out["rot_" + tag] = p.phi_xy.values
    save("ref_phase_prior.npz", out)
    print("[phase prior] ok")


def make_tutorial_flow():
    """The tutorials' two-stage flow on a two-sample data set (Tutorial_Aissa_PC9_TwoSample cells 23-46): phase fit with a
    per-batch sigma_dnu tensor, hand-over of phi_xy / nu / dnu / shape_inv exactly as tutorial cell 42 builds
    `condition_on_dict`, then the default (LRMN) velocity fit with one angular speed per condition."""
    n1, n2, seed = 20, 20, 21
    d, ad, cyc, ph, Db = build_inputs(30, 10, 1, 2, seed=9)
    s = torch.ones((2, 10, 1)); s[0] = 0.001; s[1] = 0.1
    out = {"S": ad.layers["spliced"], "U": ad.layers["unspliced"], "batch": np.array(ad.obs["batch"]).astype(str),
           "cyc_means": cyc.means.values, "cyc_stds": cyc.stds.values, "phi_xy": ph.phi_xy.values, "sd_dnu": s.numpy(),
           "n1": np.array(n1), "n2": np.array(n2), "seed": np.array(seed)}
    opt = lambda n: pyro.optim.ClippedAdam({"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)})
    pyro.clear_param_store()
    mp = vc.preprocessing.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1, σΔν=s)
    pf = vc.phase_inference_model.PhaseFitModel(mp, num_samples=4, n_per_bin=2)
    torch.manual_seed(seed)
    pf.fit(opt(n1), loss=pyro.infer.Trace_ELBO(num_particles=1), num_steps=n1, verbose=False)
    out["phase_losses"] = np.array(pf.losses)
    for a in ("phis_pyro", "fourier_coef", "fourier_coef_sd", "disp_pyro", "delta_nus"):
        out["phase_" + a] = np.asarray(getattr(pf, a))
    cycle_pyro, phase_pyro = pf.cycle_pyro, pf.phase_pyro
    cond = {"ϕxy": phase_pyro.phi_xy_tensor.T, "ν": cycle_pyro.means_tensor.T.unsqueeze(-2),
            "Δν": torch.tensor(pf.delta_nus), "shape_inv": torch.tensor(pf.disp_pyro).unsqueeze(-1)}
    spd = vc.angularspeed.AngularSpeed.trivial_prior(condition_names=["b0", "b1"], harmonics=0)
    pyro.clear_param_store()
    mv = vc.preprocessing.preprocess_for_velocity_estimation(ad, cycle_pyro, phase_pyro, spd, Db.float(), Db.float(),
                                                             n_harmonics=1, count_factor=mp.count_factor, ω_n_harmonics=0,
                                                             condition_on=cond)
    vf = vc.velocity_inference_model.VelocityFitModel(mv, condition_on=cond, num_samples=4, n_per_bin=2)
    torch.manual_seed(seed + 1)
    vf.fit(opt(n2), loss=pyro.infer.Trace_ELBO(num_particles=1), num_steps=n2, verbose=False)
    out["vel_losses"] = np.array(vf.losses)
    for a in ("phis_pyro", "fourier_coef", "disp_pyro", "log_betas", "delta_nus"):
        out["vel_" + a] = np.asarray(getattr(vf, a))
    out["vel_loc"] = pyro.param("loc").detach().numpy()
    out["vel_logβg_scales"] = pyro.param("logβg_scales").detach().numpy()
    save("ref_tutorial_flow.npz", out)
    print(f"[tutorial flow] phase {pf.losses[-1]:.3f} velocity {vf.losses[-1]:.3f}")


def make_particles(seed=11, K=3, n=12):
    """`fit(loss=Trace_ELBO(num_particles=3))` of the reference's own fit drivers (velocity_inference_model.py:79,111 passes
    the user's ELBO object into SVI): K guide draws per step, loss and gradients averaged.  -> ref_fitK3_<case>.npz"""
    for name in ("vel_mf_joint", "phase_nb", "vel_lrmn_cond"):
        c = CASES[name]
        d, ad, cyc, ph, Db = build_inputs(c["Nc"], c["Ng"], c["H"], c["nb"], seed)
        Nc = ad.n_obs
        rs = np.random.RandomState(seed + 1)
        cond = {}
        if c["kind"] == "phase":
            mp = vc.preprocessing.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=c["H"], noisemodel=c["noise"],
                                                                  with_delta_nu=c["wdn"])
            FitCls = vc.phase_inference_model.PhaseFitModel
        else:
            spd = vc.angularspeed.AngularSpeed.trivial_prior(condition_names=[f"b{i}" for i in range(c["nb"])], harmonics=c["Hw"])
            if c["Hw"] == 1:
                spd.stds.loc["nu1_cos"] = [0.05] * c["nb"]
                spd.stds.loc["nu1_sin"] = [0.05] * c["nb"]
            cf = torch.tensor(np.log(ad.layers["spliced"].sum(1) / ad.layers["spliced"].sum(1).mean())).float()[None, None, :]
            for site in c.get("cond", []):
                if site == "ϕxy":
                    cond[site] = ph.phi_xy_tensor.T + torch.tensor(0.05 * rs.randn(Nc, 2)).float()
                elif site == "ν":
                    cond[site] = cyc.means_tensor.T.unsqueeze(-2) + torch.tensor(0.05 * rs.randn(c["Ng"], 1, 2 * c["H"] + 1)).float()
                elif site == "shape_inv":
                    cond[site] = torch.tensor(rs.uniform(0.2, 1.0, (c["Ng"], 1))).float()
            mp = vc.preprocessing.preprocess_for_velocity_estimation(
                ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=c["H"], ω_n_harmonics=c["Hw"], count_factor=cf,
                noisemodel=c["noise"], with_delta_nu=c["wdn"], condition_on=cond, model_type=c.get("model_type", "lrmn"))
            FitCls = vc.velocity_inference_model.VelocityFitModel
        p64 = orc.problem_from_metaparams(mp, c["kind"], cond, dtype=torch.float64)
        p32 = p64.to(torch.float32)
        opt_args = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)}
        fitm = FitCls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
        pyro.clear_param_store()
        torch.manual_seed(seed)
        fitm.fit(pyro.optim.ClippedAdam(dict(opt_args)), loss=pyro.infer.Trace_ELBO(num_particles=K), num_steps=n, verbose=False)
        ref_par, _ = ref_params_canonical(p32)
        o_losses, o_par = orc.fit(p32, opt_args, n, seed=seed, num_particles=K)
        check(o_losses, fitm.losses, f"{name}: K={K} fit losses", 1e-4, 1e-2)
        for k in ref_par:
            check(o_par[k], ref_par[k], f"{name}: K={K} fitted {k}", 2e-3, 2e-3)
        l64s, par64f = orc.fit(p64, opt_args, n, seed=seed, num_particles=K)
        fo = problem_arrays(p32)
        fo.update({"reffit_" + k: v.numpy() for k, v in ref_par.items()})
        fo.update({"fit64_" + k: v.numpy() for k, v in par64f.items()})
        fo.update(ref_losses=np.array(fitm.losses), losses64=np.array(l64s), num_steps=np.array(n), seed=np.array(seed),
                  num_particles=np.array(K))
        for k, v in opt_args.items():
            fo["opt_" + k] = np.array(v)
        save(f"ref_fitK{K}_{name}.npz", fo)
        print(f"[fit K={K}] {name}: {n} steps, ref final loss {fitm.losses[-1]:.4f}, oracle {o_losses[-1]:.4f}")


def _case_inputs(name, seed):
    """(metaparams, condition dict, fit class, Problem64) of one CASES entry, built like make_case does."""
    c = CASES[name]
    d, ad, cyc, ph, Db = build_inputs(c["Nc"], c["Ng"], c["H"], c["nb"], seed)
    Nc = ad.n_obs
    rs = np.random.RandomState(seed + 1)
    cond = {}
    if c["kind"] == "phase":
        mp = vc.preprocessing.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=c["H"], noisemodel=c["noise"],
                                                              with_delta_nu=c["wdn"])
        FitCls = vc.phase_inference_model.PhaseFitModel
    else:
        spd = vc.angularspeed.AngularSpeed.trivial_prior(condition_names=[f"b{i}" for i in range(c["nb"])], harmonics=c["Hw"])
        if c["Hw"] == 1:
            spd.stds.loc["nu1_cos"] = [0.05] * c["nb"]
            spd.stds.loc["nu1_sin"] = [0.05] * c["nb"]
        cf = torch.tensor(np.log(ad.layers["spliced"].sum(1) / ad.layers["spliced"].sum(1).mean())).float()[None, None, :]
        for site in c.get("cond", []):
            if site == "ϕxy":
                cond[site] = ph.phi_xy_tensor.T + torch.tensor(0.05 * rs.randn(Nc, 2)).float()
            elif site == "ν":
                cond[site] = cyc.means_tensor.T.unsqueeze(-2) + torch.tensor(0.05 * rs.randn(c["Ng"], 1, 2 * c["H"] + 1)).float()
            elif site == "shape_inv":
                cond[site] = torch.tensor(rs.uniform(0.2, 1.0, (c["Ng"], 1))).float()
        mp = vc.preprocessing.preprocess_for_velocity_estimation(
            ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=c["H"], ω_n_harmonics=c["Hw"], count_factor=cf,
            noisemodel=c["noise"], with_delta_nu=c["wdn"], condition_on=cond, model_type=c.get("model_type", "lrmn"))
        FitCls = vc.velocity_inference_model.VelocityFitModel
    return mp, cond, FitCls, orc.problem_from_metaparams(mp, c["kind"], cond, dtype=torch.float64)


def make_continue(seed=11, n=10):
    """A second `fit()` WITHOUT `pyro.clear_param_store()` in between: the guides' `pyro.param(name, init)` return the stored
    values (velocity_inference_guide.py:25-43, phase_inference_guide.py:36-45), so the optimisation continues; the reference's
    own comment at velocity_inference_model.py:79 plans exactly that (two fits, another ELBO object for the second).  Two
    scenarios per case, both run by the reference's own fit drivers:
      same   the SAME optimizer and ELBO objects again: moments, step count and the decayed learning rate carry on, and the
             used ELBO object makes no extra guide pass;
      new    a NEW optimizer and a NEW Trace_ELBO object: parameters carry on, optimiser state and learning-rate schedule
             start afresh, and the fresh ELBO object makes its extra guide pass (one eps set drawn and discarded).
    The global RNG is re-seeded before the second fit (the posterior draws at the end of the first fit consume it in a way
    that is not part of the path).  -> ref_fit_continue_<case>.npz"""
    for name in ("vel_mf_joint", "phase_nb", "vel_lrmn_cond"):
        mp, cond, FitCls, p64 = _case_inputs(name, seed)
        p32 = p64.to(torch.float32)
        opt_args = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / (2 * n)), "betas": (0.80, 0.99)}
        fo = problem_arrays(p32)
        for scen in ("same", "new"):
            fitm = FitCls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
            pyro.clear_param_store()
            opt1, elbo1 = pyro.optim.ClippedAdam(dict(opt_args)), pyro.infer.Trace_ELBO(num_particles=1)
            torch.manual_seed(seed)
            fitm.fit(opt1, loss=elbo1, num_steps=n, verbose=False)
            l1 = list(fitm.losses)
            par1, _ = ref_params_canonical(p32)
            opt2, elbo2 = (opt1, elbo1) if scen == "same" else (pyro.optim.ClippedAdam(dict(opt_args)), pyro.infer.Trace_ELBO(num_particles=1))
            torch.manual_seed(seed + 1)
            fitm.fit(opt2, loss=elbo2, num_steps=n, verbose=False)              # NO clear_param_store in between
            l2 = list(fitm.losses)
            par2, _ = ref_params_canonical(p32)
            # the oracle, told the same story
            oopt = orc.ClippedAdam(opt_args)
            ol1, opar1 = orc.fit(p32, opt_args, n, seed=seed, opt=oopt)
            check(ol1, l1, f"{name}/{scen}: first fit losses", 1e-4, 1e-2)
            ol2, opar2 = orc.fit(p32, opt_args, n, seed=seed + 1, params={k: v.clone() for k, v in opar1.items()},
                                 warmup_draw=(scen == "new"), opt=(oopt if scen == "same" else None))
            check(ol2, l2, f"{name}/{scen}: second fit losses", 1e-4, 1e-2)
            for k in par2:
                check(opar2[k], par2[k], f"{name}/{scen}: fitted {k}", 2e-3, 2e-3)
            o64 = orc.ClippedAdam(opt_args)
            l64a, par64a = orc.fit(p64, opt_args, n, seed=seed, opt=o64)
            l64b, par64b = orc.fit(p64, opt_args, n, seed=seed + 1, params={k: v.clone() for k, v in par64a.items()},
                                   warmup_draw=(scen == "new"), opt=(o64 if scen == "same" else None))
            fo.update({f"{scen}_reffit1_" + k: v.numpy() for k, v in par1.items()})
            fo.update({f"{scen}_reffit2_" + k: v.numpy() for k, v in par2.items()})
            fo.update({f"{scen}_fit64_" + k: v.numpy() for k, v in par64b.items()})
            fo[f"{scen}_ref_losses1"], fo[f"{scen}_ref_losses2"] = np.array(l1), np.array(l2)
            fo[f"{scen}_losses64_1"], fo[f"{scen}_losses64_2"] = np.array(l64a), np.array(l64b)
            print(f"[continue/{scen}] {name}: {n}+{n} steps, ref losses {l1[-1]:.4f} -> {l2[0]:.4f} .. {l2[-1]:.4f}, oracle {ol2[-1]:.4f}")
        fo.update(num_steps=np.array(n), seed=np.array(seed))
        for k, v in opt_args.items():
            fo["opt_" + k] = np.array(v)
        save(f"ref_fit_continue_{name}.npz", fo)


def _finish(what):
    kind, ver = ref_loader.BACKEND
    on = f"pyro-ppl {ver}" if kind == "real" else "oracle/pyro_shim (pyro-ppl is not installed: the Pyro boundary stays restated)"
    if CHECK:
        for m in MISMATCH:
            print("MISMATCH", m)
        print(f"--check on {on}: {what}: " + ("every regenerated array agrees with the committed fixtures" if not MISMATCH
                                              else f"{len(MISMATCH)} disagreement(s)"))
        sys.exit(1 if MISMATCH else 0)
    print(f"{what} written on {on}; oracle == reference on every case")
    sys.exit(0)


if __name__ == "__main__":
    if ARGS == ["--continue"]:
        make_continue()
        _finish("continue fixtures")
    if ARGS == ["--particles"]:
        make_particles()
        _finish("particle fixtures")
    if ARGS == ["--all"]:
        make_continue()
        make_particles()
        ARGS = []
    only = ARGS
    if not only:
        make_basis()
        make_tutorial_flow()
        make_preprocess()
        make_phase_prior()
    for nm, c in CASES.items():
        if only and nm not in only:
            continue
        make_case(nm, c)
    _finish("golden fixtures")
