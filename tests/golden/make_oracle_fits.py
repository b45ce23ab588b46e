"""Converged-fit trajectories of the ORACLE (float64 and float32) for the acceptance tests of tests/test_hip_acceptance.py.

    python tests/golden/make_oracle_fits.py        (CPU only, ~6 min on 8 cores; writes tests/golden/oracle_fit_*.npz)

north_star's acceptance statement is "posterior means within 1e-3 rel of reference" after a real-length fit.  Running the
oracle for 1 500 steps at 3 000 cells x 200 genes inside the GPU test suite would take minutes per case, so its
trajectories are computed here once and committed as data: losses and final unconstrained parameters of `orc.fit` in float64
(the checker) and in float32 (= the arithmetic the reference runs in; its distance from float64 is the yardstick for what
"equal" can mean after 1 500 Adam steps), on the deterministic synthetic workloads of velocycle_amd.workloads (CPU
generator, seeds below) with the host eps stream of `seed` (the reference's RNG order, oracle.draw_eps).  A digest of the
inputs is stored so that the test can prove it rebuilt the same problem."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import velocycle_oracle as orc      # noqa: E402
from tests import helpers as H                   # noqa: E402

N_STEPS = 1500
OPT = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / N_STEPS), "betas": (0.80, 0.99)}      # tutorial cells 27 / 43 / 56
# simulated data sets (velocycle_amd.simulate.simulate_counts: the recipe of reference utils.py:508-584), stored with the
# fixtures because torch's CPU sampling / transcendental kernels are not bit-reproducible across hosts
DATA = {"A": dict(Nc=3000, Ng=200, omegas=(0.4,), seed=5), "B": dict(Nc=1500, Ng=200, omegas=(0.4, 0.3), seed=6)}
# name -> (data set, workload kwargs, eps seed)
CASES = {
    "vjoint_3000x200": ("A", dict(mode="vjoint", n_conditions=1, Hw=1), 11),
    "vcond_3000x200": ("A", dict(mode="vcond", n_conditions=1, Hw=1), 12),
    "vjoint2_1500x200": ("B", dict(mode="vjoint", n_conditions=2, Hw=0), 13),   # two samples, omega 0.4 / 0.3
    "phase_3000x200": ("A", dict(), 14),
}


def data_path(key):
    return os.path.join(HERE, f"oracle_fit_data_{key}.npz")


def load_sim(key):
    """The stored simulation as the dict `simulate_counts` returns (counts back to float32)."""
    z = np.load(data_path(key))
    sim = {k: torch.from_numpy(z[k].astype(np.float32) if k in ("S", "U") else z[k]) for k in z.files if k != "omegas"}
    sim["omegas"] = tuple(float(x) for x in z["omegas"])
    return sim


def make_spec(name):
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    key, kw, _ = CASES[name]
    sim = load_sim(key)
    Nc, Ng = DATA[key]["Nc"], DATA[key]["Ng"]
    if name.startswith("phase"):
        return make_phase_spec(Nc, Ng, sim=sim)
    return make_velocity_spec(Nc, Ng, sim=sim, **kw)


def digest(spec) -> str:
    h = hashlib.sha256()
    for t in (spec.S, spec.U):            # integer counts: exact on every host (the priors derived from them are float ops)
        if t is not None:
            h.update(np.ascontiguousarray(t.detach().cpu().float().numpy()).tobytes())
    return h.hexdigest()


def main():
    from velocycle_amd.simulate import simulate_counts
    for key, kw in DATA.items():
        sim = simulate_counts(**kw)
        assert float(sim["S"].max()) < 65536 and float(sim["U"].max()) < 65536
        out = {k: (v.numpy().astype(np.uint16) if k in ("S", "U") else v.numpy()) for k, v in sim.items() if torch.is_tensor(v)}
        out["omegas"] = np.array(sim["omegas"])
        np.savez_compressed(data_path(key), **out)
    for name, (key, kw, seed) in CASES.items():
        spec = make_spec(name)
        out = {"digest": digest(spec), "seed": seed, "n_steps": N_STEPS, "opt_lr": OPT["lr"], "opt_lrd": OPT["lrd"],
               "opt_betas": np.array(OPT["betas"])}
        for tag, dt in (("64", torch.float64), ("32", torch.float32)):
            t0 = time.time()
            losses, par = orc.fit(H.problem_from_spec(spec, dt), OPT, N_STEPS, seed=seed)
            out["loss" + tag] = np.array(losses, dtype=np.float64)
            for k, v in par.items():
                out[f"par{tag}_{k}"] = v.detach().double().numpy()
            print(name, tag, f"{time.time() - t0:.0f}s", "loss", losses[0], "->", losses[-1], flush=True)
        np.savez_compressed(os.path.join(HERE, f"oracle_fit_{name}.npz"), **out)


def reference_metaparams(name, vc):
    """The workload of `make_spec(name)` handed to the REFERENCE's own preprocess_for_* through its own containers (tutorial
    cells 19-21, 38-41): returns (metaparams, condition dict, fit class, kind).  main_reference() proves field by field that
    the reference's container describes the same problem as the spec the oracle / the HIP engine run on."""
    from velocycle_amd.anndata_lite import AnnDataLite
    key, kw, _ = CASES[name]
    sp = make_spec(name)
    ad = AnnDataLite(sp.S.t().numpy(), (sp.U if sp.U is not None else sp.S).t().numpy())
    batch = sp.truth["batch"]
    ad.obs["batch"] = [f"s{int(b)}" for b in batch]
    genes = list(ad.var.index)
    cyc = vc.cycle.Cycle.trivial_prior(gene_names=genes, harmonics=sp.H)
    cyc.set_means(sp.mu_nu.T.double().numpy())
    cyc.set_stds(sp.sd_nu.T.double().numpy())
    ph = vc.phases.Phases.from_array(sp.phixy_prior.T.double().numpy(), cell_names=list(ad.obs.index))
    Db = vc.preprocessing.make_design_matrix(ad, ids="batch")
    if name.startswith("phase"):
        mp = vc.preprocessing.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=sp.H, with_delta_nu=False)
        return mp, {}, vc.phase_inference_model.PhaseFitModel, "phase"
    nb = sp.Nx
    spd = vc.angularspeed.AngularSpeed.trivial_prior(condition_names=[f"s{i}" for i in range(nb)], harmonics=sp.Hw)
    if sp.Hw == 1:
        spd.stds.loc["nu1_cos"] = [0.05] * nb
        spd.stds.loc["nu1_sin"] = [0.05] * nb
    cond = {}
    if kw["mode"].startswith("vcond"):
        c = sp.condition_on
        cond = {"ϕxy": c["ϕxy"].float(), "ν": c["ν"].float().unsqueeze(-2), "shape_inv": c["shape_inv"].float().unsqueeze(-1)}
        if "Δν" in c:
            cond["Δν"] = c["Δν"].float().reshape(nb, 1, 1, sp.Ng, 1)
    cf = sp.count_factor.float()[None, None, :]
    mp = vc.preprocessing.preprocess_for_velocity_estimation(
        ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=sp.H, ω_n_harmonics=sp.Hw, count_factor=cf,
        with_delta_nu=sp.with_delta_nu, condition_on=cond, model_type=("lrmn" if sp.guide == "lrmn" else "normal"))
    return mp, cond, vc.velocity_inference_model.VelocityFitModel, "velocity"


def main_reference():
    """`python tests/golden/make_oracle_fits.py --reference` (build container only: needs /root/reference): the SAME 1 500-step fits
    run by the reference's own `PhaseFitModel.fit` / `VelocityFitModel.fit` (velocity_inference_model.py:111-187,
    phase_inference_model.py:162-201; unmodified files from build/lib on oracle/pyro_shim), float32 as the reference computes,
    `torch.manual_seed(seed)` = the eps stream of the oracle fits -> ref_fitlong_<case>.npz (losses + fitted unconstrained
    parameters in canonical shapes).  Aborts unless the reference's container and the workload spec describe the same problem."""
    from oracle import ref_loader
    vc = ref_loader.load_reference("real" if "--real-pyro" in sys.argv[1:] else "auto")      # real pyro-ppl 1.8.x when installed, else the shim
    import pyro
    print("reference on", ref_loader.BACKEND, flush=True)
    from tests.golden.make_golden import CANON
    for name, (key, kw, seed) in CASES.items():
        mp, cond, FitCls, kind = reference_metaparams(name, vc)
        spec = make_spec(name)
        pr = orc.problem_from_metaparams(mp, kind, cond, dtype=torch.float32)
        ps = H.problem_from_spec(spec, torch.float32)
        for k, v in ps.__dict__.items():
            w = getattr(pr, k)
            if isinstance(v, torch.Tensor):
                if not torch.allclose(w.reshape(v.shape), v, rtol=1e-6, atol=1e-6):
                    raise SystemExit(f"{name}: the reference's container differs from the workload spec in {k}: {float((w.reshape(v.shape) - v).abs().max())}")
            elif k == "condition_on":
                assert set(v) == set(w), (name, set(v), set(w))
                for a in v:
                    assert torch.allclose(pr.cond(a), ps.cond(a), rtol=1e-6, atol=1e-6), (name, a)
            elif k == "sd_dnu":
                assert kind == "phase" or float(w) == float(v), (name, k, v, w)
            elif isinstance(v, float):
                assert abs(float(w) - v) <= 1e-6 * max(1.0, abs(v)), (name, k, v, w)
            else:
                assert w == v, (name, k, v, w)
        t0 = time.time()
        fitm = FitCls(mp, condition_on=cond, num_samples=4, n_per_bin=2)
        pyro.clear_param_store()
        torch.manual_seed(seed)
        fitm.fit(pyro.optim.ClippedAdam(dict(OPT)), loss=pyro.infer.Trace_ELBO(num_particles=1), num_steps=N_STEPS, verbose=False)
        store = pyro.get_param_store()
        out = {"digest": digest(spec), "seed": seed, "n_steps": N_STEPS, "ref_losses": np.array(fitm.losses, dtype=np.float64)}
        uncon = ref_loader.unconstrained_params(store)
        for pn in store.keys():
            out["reffit_" + pn] = uncon[pn].detach().reshape(CANON[pn](pr)).double().numpy()
        np.savez_compressed(os.path.join(HERE, f"ref_fitlong_{name}.npz"), **out)
        z = np.load(os.path.join(HERE, f"oracle_fit_{name}.npz"))
        l32, l64, lr = z["loss32"], z["loss64"], out["ref_losses"]
        print(name, f"{time.time() - t0:.0f}s", "reference loss", lr[0], "->", lr[-1], "| vs oracle32: first 5", np.abs(lr[:5] / l32[:5] - 1).max(),
              "max", np.abs(lr / l32 - 1).max(), "median", np.median(np.abs(lr / l32 - 1)), "| oracle32 vs 64 max", np.abs(l32 / l64 - 1).max(), flush=True)


if __name__ == "__main__":
    if "--reference" in sys.argv[1:]:
        main_reference()
    else:
        main()
