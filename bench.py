#!/usr/bin/env python
"""Benchmark of the SVI hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one full SVI step of velocity inference on the BASELINE.json workload (synthetic
50k cells x 2k genes): guide sampling -> ELBO + reparameterised gradient (HIP) -> [all-reduce over
cell shards when N > 1] -> ClippedAdam update.  N > 1: launched by torch.distributed.run, one rank per
GPU, cells sharded contiguously (strong scaling: the problem is fixed, value = steps/s of the job).

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline      HIP-event timing of the likelihood kernel against the HBM roof (ALGORITHMIC bytes:
                fp32 count matrices read once, 8*Ng*Nc for the joint workload),
  cpu_baseline  the oracle restatement (op-by-op torch fp32 + autograd + ClippedAdam) timed on this
                host's cores on a bounded sample of the same workload (rank 0, N=1 only),
  modes         steps/s of the tutorial flow (velocity conditioned on the phase fit, default LRMN guide) and of
                phase_inference at the same size, next to the headline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def baseline_metric():
    """BASELINE.json's metric string (the file travels with the repo); literal fallback if it is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "SVI steps/sec + ELBO-match, velocity_inference 50k cells\u00d72k genes, 1/2/4/8 GPU"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--mode", default="vjoint", choices=["vjoint", "vcond", "vcond_mf", "phase"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-modes", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=5000)
    return ap.parse_args()


def time_steps(run, steps, warmup, dist_on, device):
    import torch.distributed as dist
    run.run_perf(warmup, sync=True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run.run_perf(steps, sync=False)
    torch.cuda.synchronize(device)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def kernel_roofline(engine, run, steps):
    """Average duration of the likelihood kernel over `steps` eager SVI steps, from hipEvents recorded
    by the library on the launch stream around that kernel only."""
    saved_graph, saved_flag = run._graph, run.use_graph
    run._graph, run.use_graph = None, False
    engine.set_timing(True)
    run.run_perf(steps, sync=True)
    ms, n = engine.get_timing()
    engine.set_timing(False)
    run._graph, run.use_graph = saved_graph, saved_flag
    avg_s = ms / max(n, 1) * 1e-3
    st = engine.stats
    achieved = st["algorithmic_bytes"] / avg_s / 1e9
    traffic, traffic_src = None, None
    try:     # PMC numbers cannot be collected from inside the process: they come from the committed rocprofv3 passes
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
        ent = tj["kernels"].get(st["main_kernel"])
        if ent and tj.get("workload") == f"{engine.spec.Nc}x{engine.spec.Ng}" and engine.world_size == 1:
            traffic, traffic_src = int(ent["traffic_bytes"]), "profiles/latest_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per gfx950 note)"
    except Exception:
        pass
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": st["main_kernel"], "kernel_avg_us": round(avg_s * 1e6, 2), "launches": int(n),
            "algorithmic_bytes_per_launch": int(st["algorithmic_bytes"]),
            "streamed_bytes_per_launch": int(st["streamed_bytes"]),
            "method": "hipEvents around the kernel over eager SVI steps run right after the timed region"}


def cpu_baseline(args, mode, device=None):
    """Oracle restatement timed on the host: same workload, first `cpu_sample_cells` cells, scaled."""
    from oracle import velocycle_oracle as orc
    from tests import helpers as H
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    nsample = min(args.cpu_sample_cells, args.cells)
    spec = (make_phase_spec(nsample, args.genes, seed=0, device="cpu") if mode == "phase"
            else make_velocity_spec(nsample, args.genes, mode, 1, 1, seed=0, device="cpu"))
    kw = {}
    for k, v in spec.__dict__.items():
        if k == "truth":
            continue
        kw[k] = v.contiguous() if isinstance(v, torch.Tensor) else v
    p = orc.Problem(**kw)
    gen = torch.Generator().manual_seed(0)
    first = orc.draw_eps(p, gen)
    params = orc.init_params(p, first.get("_cov_factor_draw"))
    opt = orc.ClippedAdam({"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)})

    # ELBO-match (the other half of BASELINE.json's metric): the HIP path and the oracle on identical (params, eps)
    eps0 = orc.draw_eps(p, gen)
    loss_cpu, _, _, _ = orc.loss_and_grads(p, params, eps0)
    elbo_match = None
    if device is not None:
        from velocycle_amd.engine import HipEngine
        eng = HipEngine(spec, device=device)
        eng.set_params({k: v.float() for k, v in params.items()})
        eng.elbo_grad(eps=eng.pack_eps({k: v.float() for k, v in eps0.items() if not k.startswith("_")}))
        torch.cuda.synchronize(device)
        loss_hip = eng.loss()
        eng.close()
        elbo_match = {"loss_hip": loss_hip, "loss_cpu_port": float(loss_cpu),
                      "rel_err": abs(loss_hip - float(loss_cpu)) / abs(float(loss_cpu)),
                      "note": "one ELBO evaluation on the CPU sample with identical params and eps; the port runs in "
                              "float32, so this bounds both sides' rounding (tests compare against float64 at 1e-5)"}

    def one():
        nonlocal params
        eps = orc.draw_eps(p, gen)
        _, grads, _, _ = orc.loss_and_grads(p, params, eps)
        params = opt.step(params, grads)
    one()                                   # warm-up
    # the op-by-op torch path does not scale to every hardware thread of a big host: pick the thread count
    # that is fastest on this box (short sweep), then time with it
    default_nt = torch.get_num_threads()
    sweep = {}
    for nt in sorted({8, 16, 32, 64, default_nt}):
        if nt > (os.cpu_count() or nt):
            continue
        torch.set_num_threads(nt)
        one()
        t0 = time.perf_counter()
        one()
        sweep[nt] = time.perf_counter() - t0
    best_nt = min(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)
    t0 = time.perf_counter()
    n = 0
    while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 50):
        one()
        n += 1
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_nt)
    sps_sample = n / dt
    scale = nsample / args.cells
    return {"value": round(sps_sample * scale, 4), "unit": "SVI steps/s", "cores": best_nt,
            "kind": "port", "elbo_match": elbo_match,
            "sample": f"oracle (op-by-op torch fp32 + autograd + ClippedAdam) on the first {nsample} of "
                      f"{args.cells} cells x {args.genes} genes, {n} steps in {dt:.1f}s = {sps_sample:.3f} steps/s, "
                      f"scaled by {scale:.3f} (cost is linear in cells); {best_nt} torch threads = fastest of sweep "
                      f"{ {k: round(v, 2) for k, v in sweep.items()} } s/step; host cpu_count={os.cpu_count()}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and dist_on:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and not dist_on:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    # test hook (tests/test_hip_multiproc.py): VC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and exchanges through
    # gloo, so that the N > 1 path can be exercised end to end on a 1-GPU box.  Never set in a measured run.
    one_device = os.environ.get("VC_BENCH_ONE_DEVICE", "0") == "1"
    device = torch.device("cuda:0" if one_device else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec

    optim = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / 10000), "betas": (0.80, 0.99)}

    def build(mode):
        if mode == "phase":
            spec = make_phase_spec(args.cells, args.genes, seed=0, device=device)
        else:
            spec = make_velocity_spec(args.cells, args.genes, mode, 1, 1, seed=0, device=device)
        eng = HipEngine(spec, device=device, rank=rank, world_size=world)
        # N > 1: eager launches by default.  Capturing the RCCL all-reduce into the hipGraph works with a 1-rank
        # group on the 1-GPU box (tests/test_hip_svi.py) but cannot be exercised across ranks there, and a step of
        # 6 asynchronous launches stays GPU-bound anyway; VC_BENCH_DIST_GRAPH=1 opts in.
        graph = None
        if args.no_graph or (dist_on and os.environ.get("VC_BENCH_DIST_GRAPH", "0") != "1"):
            graph = False
        run = SVIRunner(eng, optim, mode="perf", seed=0, use_graph=graph)
        return spec, eng, run

    spec, eng, run = build(args.mode)
    dt = time_steps(run, args.steps, args.warmup, dist_on, device)
    sps = args.steps / dt
    losses = run.perf_losses()
    roof = kernel_roofline(eng, run, min(args.steps, 100))
    if dist_on:
        roof["note"] = f"per-rank kernel on {eng.Nc_local} of {args.cells} cells"
    out = {
        "metric": baseline_metric() if args.mode != "phase" else "SVI steps/sec, phase_inference",
        "value": round(sps, 2), "unit": "SVI steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic {args.cells} cells x {args.genes} genes " + ("phase_inference" if args.mode == "phase" else "velocity_inference") + ", "
                               + {"vjoint": "mean-field guide, nothing conditioned (every gradient)",
                                  "vcond": "tutorial flow: LRMN guide conditioned on phixy, nu, shape_inv",
                                  "vcond_mf": "mean-field guide conditioned on phixy, nu, shape_inv",
                                  "phase": "spliced matrix only, mean-field guide, nothing conditioned"}[args.mode]
                               + ", NegativeBinomial noise, H=1, Hw=1",
                   "cells": args.cells, "genes": args.genes, "mode": args.mode,
                   "parallelism": f"cells sharded over {world} GPU(s), one all-reduce of gene-level gradients per step",
                   "step": ("Philox eps -> ELBO+grad (HIP kernels) -> " + (("gloo (test hook) " if one_device else "RCCL ") + "all-reduce -> " if dist_on else "")
                            + f"ClippedAdam ({run.adam_impl}), " + ("hipGraph replay" if run.use_graph else "eager launches"))},
        "roofline": roof,
        "loss_first_last": [losses[0], losses[-1]],
    }
    extra = {}
    if not args.no_extra_modes and not dist_on:
        del run, eng, spec
        torch.cuda.empty_cache()
        for m in [x for x in ("vcond", "phase", "vjoint") if x != args.mode][:2]:
            s2, e2, r2 = build(m)
            dt2 = time_steps(r2, args.steps, args.warmup, False, device)
            rf = kernel_roofline(e2, r2, min(args.steps, 100))
            extra[m] = {"steps_per_s": round(args.steps / dt2, 2), "kernel": rf["kernel"],
                        "kernel_avg_us": rf["kernel_avg_us"], "hbm_achieved_GBs": rf["achieved"],
                        "hbm_frac": rf["frac"]}
            del s2, e2, r2
            torch.cuda.empty_cache()
        out["modes"] = extra
    if rank == 0 and not dist_on and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.mode, device)
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))


if __name__ == "__main__":
    main()
