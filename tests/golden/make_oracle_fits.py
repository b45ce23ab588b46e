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


if __name__ == "__main__":
    main()
