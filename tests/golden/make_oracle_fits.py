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
# name -> (workload kwargs, eps seed)
CASES = {
    "vjoint_3000x200": (dict(Nc=3000, Ng=200, mode="vjoint", n_conditions=1, Hw=1, seed=5), 11),
    "vcond_3000x200": (dict(Nc=3000, Ng=200, mode="vcond", n_conditions=1, Hw=1, seed=5), 12),
    "vjoint2_1500x200": (dict(Nc=1500, Ng=200, mode="vjoint", n_conditions=2, Hw=0, seed=6), 13),   # two samples, omega 0.4 / 0.3
    "phase_3000x200": (dict(Nc=3000, Ng=200, seed=5), 14),
}


def make_spec(name):
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    kw, _ = CASES[name]
    return make_phase_spec(**kw) if name.startswith("phase") else make_velocity_spec(**kw)


def digest(spec) -> str:
    h = hashlib.sha256()
    for t in (spec.S, spec.U, spec.count_factor, spec.phixy_prior, spec.mu_nu, spec.sd_nu):
        if t is not None:
            h.update(np.ascontiguousarray(t.detach().cpu().float().numpy()).tobytes())
    return h.hexdigest()


def main():
    for name, (kw, seed) in CASES.items():
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
