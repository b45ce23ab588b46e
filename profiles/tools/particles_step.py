#!/usr/bin/env python
"""us per SVI step of Trace_ELBO(num_particles = K) from one C call (vc_svi_run_particles, batched layout: K + 3 launches) at
50 000 cells x 2 000 genes, with the small kernels compiled for the configuration's signature (default) and with the run-time-flag
kernels (Tuning(no_tail_spec=True)), alternated; K_main from hipEvents.   python profiles/tools/particles_step.py [mode] [K]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from velocycle_amd.engine import HipEngine  # noqa: E402
from velocycle_amd.svi import SVIRunner  # noqa: E402
from velocycle_amd.tuning import Tuning  # noqa: E402
from velocycle_amd.workloads import make_phase_spec, make_velocity_spec  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "vjoint"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
OPT = {"lr": 0.03, "lrd": 0.9999, "betas": (0.8, 0.99)}
spec = make_phase_spec(50000, 2000, seed=0, device=dev) if mode == "phase" else make_velocity_spec(50000, 2000, mode, 1, 1, seed=0, device=dev)


def timed(run, n=200, reps=5):
    run.run_perf(40, sync=True)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.run_perf(n, sync=True)
        ts.append((time.perf_counter() - t0) / n * 1e6)
    return round(sorted(ts)[len(ts) // 2], 2)


for rep in range(2):
    for name, tun in (("compiled signature", Tuning()), ("run-time flags", Tuning(no_tail_spec=True))):
        eng = HipEngine(spec, device=dev, tuning=tun)
        run = SVIRunner(eng, OPT, mode="perf", seed=0, num_particles=K)
        t = timed(run)
        eng.set_timing(True)
        run.run_perf(100, sync=True)
        ms, k = eng.get_timing()
        eng.set_timing(False)
        km = round(ms / max(k, 1) * 1e3, 2)
        print(json.dumps({"mode": mode, "K": K, "small_kernels": eng.stats["tail_spec_name"], "us_per_step": t, "K_main_us": km,
                          "us_per_step_minus_K_x_K_main": round(t - K * km, 1)}), flush=True)
        eng.close()
        del run, eng
