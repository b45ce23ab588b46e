#!/usr/bin/env python
"""Per-kernel durations of the single-rank fused step and of the sharded fused step (no exchange between its phases) at one
shard size, for `rocprofv3 --kernel-trace --stats -- python3 profiles/tools/sharded_kernels.py CELLS [mode]`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from velocycle_amd.engine import HipEngine  # noqa: E402
from velocycle_amd.tuning import Tuning
from velocycle_amd.svi import SVIRunner  # noqa: E402
from velocycle_amd.workloads import make_velocity_spec  # noqa: E402

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
mode = sys.argv[2] if len(sys.argv) > 2 else "vjoint"
dev = torch.device("cuda:0")
spec = make_velocity_spec(nc, 2000, mode, 1, 1, seed=0, device=dev)
OPT = {"lr": 0.03, "lrd": 0.9999, "betas": (0.8, 0.99)}
for kw in (dict(adam_impl="fused3"), dict(adam_impl="sharded", exchange="none", force_reduce=True)):
    eng = HipEngine(spec, device=dev, tuning=Tuning.from_env())
    run = SVIRunner(eng, OPT, mode="perf", seed=0, use_graph=False, **kw)
    run.run_perf(1500, sync=True)
    eng.close()
