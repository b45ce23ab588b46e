#!/usr/bin/env python
"""Wall time of the tutorial-shaped flow through the drop-in entry points, stage by stage (one MI355X):
preprocess_for_phase_estimation -> PhaseFitModel.fit (+ posterior) -> hand-over -> preprocess_for_velocity_estimation ->
VelocityFitModel.fit (LRMN guide conditioned on the phase fit, + posterior sampling in bins).

  python profiles/tools/fit_wall_time.py [cells] [genes] [steps] [num_samples] [sparse]

Everything outside the SVI steps is host-visible latency the reference also pays (it runs it on the CPU); this shows
where the flow spends its time once the steps themselves take ~0.1 ms."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import torch  # noqa: E402

from velocycle_amd import containers as C, preprocessing as P  # noqa: E402
from velocycle_amd.anndata_lite import AnnDataLite  # noqa: E402
from velocycle_amd.fit_models import PhaseFitModel, VelocityFitModel  # noqa: E402
from velocycle_amd.optim import ClippedAdam  # noqa: E402
from velocycle_amd.workloads import make_velocity_spec  # noqa: E402

Nc = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
Ng = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
nsamp = int(sys.argv[4]) if len(sys.argv) > 4 else 500
sparse = len(sys.argv) > 5 and sys.argv[5] == "sparse"
T = {}


class clock:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        torch.cuda.synchronize()
        self.t = time.perf_counter()

    def __exit__(self, *a):
        torch.cuda.synchronize()
        T[self.name] = round(time.perf_counter() - self.t, 3)


with clock("synthetic_data"):
    spec = make_velocity_spec(Nc, Ng, "vjoint", 1, 0, seed=0)
    S, U = spec.S.t().contiguous().numpy(), spec.U.t().contiguous().numpy()          # (cells, genes), like AnnData layers
    ad = AnnDataLite(sp.csr_matrix(S) if sparse else S, sp.csr_matrix(U) if sparse else U)
    cyc = C.Cycle.from_array(spec.mu_nu.T.numpy(), spec.sd_nu.T.numpy(), list(ad.var.index))
    ph = C.Phases.from_array(spec.phixy_prior.T.numpy(), cell_names=list(ad.obs.index))
opt = lambda n: ClippedAdam({"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)})
with clock("preprocess_phase"):
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, torch.ones(Nc, 1), n_harmonics=1, with_delta_nu=False)
with clock("phase_fit_total"):
    pf = PhaseFitModel(mp, num_samples=nsamp, n_per_bin=50)
    t0 = time.perf_counter()
    pf.fit(opt(steps), num_steps=steps, verbose=False, seed=1)
T["phase_fit_call"] = round(time.perf_counter() - t0, 3)
with clock("handover_and_preprocess_velocity"):
    cond = {"ϕxy": pf.phase_pyro.phi_xy_tensor.T, "ν": pf.cycle_pyro.means_tensor.T.unsqueeze(-2),
            "shape_inv": torch.tensor(pf.disp_pyro).unsqueeze(-1)}
    spd = C.AngularSpeed.trivial_prior(condition_names=["all"], harmonics=0)
    from velocycle_amd import pyro_compat as pyro
    pyro.clear_param_store()                  # as the tutorials do between the stages (a fit() continues from the store)
    mv = P.preprocess_for_velocity_estimation(ad, pf.cycle_pyro, pf.phase_pyro, spd, torch.ones(Nc, 1), torch.ones(Nc, 1),
                                              n_harmonics=1, count_factor=mp.count_factor, ω_n_harmonics=0, condition_on=cond,
                                              with_delta_nu=False)       # as the one-sample tutorial's cell does
with clock("velocity_fit_total"):
    vf = VelocityFitModel(mv, condition_on=cond, num_samples=nsamp, n_per_bin=50)
    vf.fit(opt(steps), num_steps=steps, verbose=False, seed=2)
T["kernel"] = vf.engine.stats["main_kernel"]
T["steps"] = steps
T["cells_genes"] = [Nc, Ng]
T["num_samples"] = nsamp
T["sparse_layers"] = sparse
for name, f in (("phase", pf), ("velocity", vf)):
    tm = getattr(f, "timings", None)
    if tm:
        T[name + "_stages"] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tm.items()}
print(json.dumps(T, ensure_ascii=False))
