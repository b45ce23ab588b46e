#!/usr/bin/env python
"""us per SVI step at the per-rank shard sizes of the 50k x 2k problem (N = 8, 4, 2, 1 ranks -> 6 250 ... 50 000 cells),
measured on ONE MI355X: what a rank spends per step before any inter-GPU latency.

  python profiles/tools/step_time_vs_shard.py [mode] [--nccl]

Columns: the unfused single-rank step (K_pre, K_main, K_post, K_fin+Adam: 4 launches, round 1), the fused step
(K_main, K_tail, K_omega: 3 launches), both replayed from a hipGraph; with --nccl also the multi-rank launch sequences on a
1-rank RCCL group: round 2's (vc_elbo_grad + all-reduce + optimiser kernel: 5 launches + the collective) and round 3's
sharded fused step (K_main -> phase A -> all-reduce -> phase B: 3 launches + the collective), graph and eager, and the
sharded step with NO exchange between its phases (what the two extra seams cost by themselves).  K_main's own duration
(hipEvents, eager) is printed next to them, so that `step - K_main` = the fixed per-step cost.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from velocycle_amd.engine import HipEngine  # noqa: E402
from velocycle_amd.tuning import Tuning
from velocycle_amd.svi import SVIRunner  # noqa: E402
from velocycle_amd.workloads import make_velocity_spec  # noqa: E402

mode = next((a for a in sys.argv[1:] if not a.startswith("--")), "vjoint")
with_nccl = "--nccl" in sys.argv
quick = "--quick" in sys.argv          # A/B runs: the eager fused step, K_main and the sharded step only, at 6 250 / 12 500 / 50 000 cells
dev = torch.device("cuda:0")
OPT = {"lr": 0.03, "lrd": 0.9999, "betas": (0.8, 0.99)}
if with_nccl:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)


def timed(run, n=2000, reps=5):
    run.run_perf(200, sync=True)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.run_perf(n, sync=True)
        ts.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(ts)[len(ts) // 2]


rows = []
for n_ranks, nc in ((8, 6250), (4, 12500), (2, 25000), (1, 50000)):
    spec = make_velocity_spec(nc, 2000, mode, 1, 1, seed=0, device=dev)
    row = {"ranks": n_ranks, "cells": nc}
    variants = [("unfused_4_launches", dict(adam_impl="fused", use_graph=True)),
                ("unfused_4_launches_eager", dict(adam_impl="fused", use_graph=False)),
                ("fused_3_launches", dict(adam_impl="fused3", use_graph=True)),
                ("fused_3_launches_eager", dict(adam_impl="fused3", use_graph=False))]
    if with_nccl:
        variants += [("multirank_seq_graph", dict(adam_impl="hip", use_graph=True, force_reduce=True)),
                     ("multirank_seq_eager", dict(adam_impl="hip", use_graph=False, force_reduce=True)),
                     ("sharded_fused_graph", dict(adam_impl="sharded", use_graph=True, force_reduce=True, exchange="torch")),
                     ("sharded_fused_eager", dict(adam_impl="sharded", use_graph=False, force_reduce=True, exchange="torch")),
                     ("sharded_fused_engine_rccl", dict(adam_impl="sharded", use_graph=False, force_reduce=True, exchange="engine")),
                     ("sharded_fused_no_exchange", dict(adam_impl="sharded", use_graph=False, force_reduce=True, exchange="none")),
                     # round 6: the one-shot peer-to-peer exchange on this one rank (own region: publish, wait, sum of one slot) --
                     # folded into phase B (default) and as the launch of its own of rounds 3-5
                     ("sharded_fused_p2p_folded", dict(adam_impl="sharded", use_graph=False, force_reduce=True, exchange="p2p", _tun=dict(p2p_fold=True))),
                     ("sharded_fused_p2p_separate", dict(adam_impl="sharded", use_graph=False, force_reduce=True, exchange="p2p", _tun=dict(p2p_fold=False))),
                     # ... and phases A and B in ONE launch with the exchange at block granularity (K_main + one launch per step)
                     ("sharded_p2p_one_launch", dict(adam_impl="sharded", use_graph=False, force_reduce=True, exchange="p2p", _tun=dict(p2p_one_launch=True)))]
    if quick:
        if nc == 25000:
            continue
        keep = ("fused_3_launches", "fused_3_launches_eager", "sharded_fused_engine_rccl", "sharded_fused_no_exchange",
                "sharded_fused_p2p_folded", "sharded_fused_p2p_separate", "sharded_p2p_one_launch")
        variants = [v for v in variants if v[0] in keep]
    for name, kw in variants:
        kw = dict(kw)
        eng = HipEngine(spec, device=dev, tuning=Tuning.from_env().replace(**kw.pop("_tun", {})))
        run = SVIRunner(eng, OPT, mode="perf", seed=0, **kw)
        row[name] = round(timed(run), 2)
        if name == "fused_3_launches":
            run._graph, run.use_graph = None, False
            eng.set_timing(True)
            run.run_perf(200, sync=True)
            ms, k = eng.get_timing()
            eng.set_timing(False)
            row["K_main"] = round(ms / max(k, 1) * 1e3, 2)
        eng.close()
        del run, eng
    rows.append(row)
    print(json.dumps(row), flush=True)
b = {r["ranks"]: r for r in rows}
for name in ("unfused_4_launches", "fused_3_launches"):
    if quick:
        break
    print(f"{name}: strong-scaling bound of one step at 8 ranks (before any all-reduce latency): "
          f"{b[1][name] / b[8][name]:.2f}x")
if with_nccl:
    dist.destroy_process_group()
