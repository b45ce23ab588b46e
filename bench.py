#!/usr/bin/env python
"""Benchmark of the SVI hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one full SVI step of velocity inference on the BASELINE.json workload (synthetic
50k cells x 2k genes): guide sampling -> ELBO + reparameterised gradient (HIP) -> [all-reduce over
cell shards when N > 1] -> ClippedAdam update.  N > 1: launched by torch.distributed.run, one rank per
GPU, cells sharded contiguously (strong scaling: the problem is fixed, value = steps/s of the job).

The timed region of --steps K steps (after --warmup W untimed ones), bracketed by barrier + synchronize and
maximised over ranks, is REPEATED (--repeats, default 9): `value` / `ms_per_step` are the MEDIAN repeat, `repeat_ms`
lists every repeat (the first ones show a clock that has not ramped yet: K = 20 steps are 4 ms of GPU work),
`device_clock_mhz` is the shader clock measured on the device right after the timed region.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline      HIP-event timing of the likelihood kernel (>= 100 launches) against the HBM roof (ALGORITHMIC
                bytes: fp32 count matrices read once, 8*Ng*Nc for the joint workload) and, under `valu`, against the
                kernel's own arithmetic bound (static VALU count of the cell loop x measured issue cost),
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


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from: the key that ties a committed PMC
    pass (profiles/latest_traffic.json) to the code it was collected on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "velocycle_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")) or fn == "Makefile":
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cells", type=int, default=50000)
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--mode", default="vjoint", choices=["vjoint", "vcond", "vcond_mf", "phase"])
    ap.add_argument("--conditions", type=int, default=1,
                    help="velocity modes: samples / conditions (Nx = Nb = this; > 1 adds the per-batch offsets Δν); the cells "
                         "are split evenly over them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-modes", action="store_true")
    ap.add_argument("--no-loss-every-demo", action="store_true",
                    help="skip the opt-in loss_every=10 measurement reported under modes.vcond.opt_in_loss_every_10")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=10000)
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of --steps steps each, the median is reported; 0 = max(9, ceil(1000 / steps)): "
                         "short regions are repeated until ~1000 steps are timed (the sustained clock under this load takes ~30 ms "
                         "of GPU work to ramp, a 20-step region is 3 ms)")
    ap.add_argument("--roofline-launches", type=int, default=100)
    ap.add_argument("--no-weak", action="store_true",
                    help="N > 1: skip the weak-scaling block (--cells cells PER RANK, same kernels) that follows the strong run")
    return ap.parse_args()


def time_steps(run, steps, warmup, dist_on, device, repeats=1):
    """`warmup` untimed steps, then `repeats` timed regions of exactly `steps` steps each; every region is bracketed by
    barrier + synchronize on both sides and its wall time is the max over ranks.  Returns the list of region times."""
    import torch.distributed as dist
    run.run_perf(warmup, sync=True)
    out = []
    for _ in range(max(1, repeats)):
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
        out.append(dt)
    return out


def median(xs):
    ys = sorted(xs)
    n = len(ys)
    return ys[n // 2] if n % 2 else 0.5 * (ys[n // 2 - 1] + ys[n // 2])


def valu_bound(engine, kernel_avg_s, clock_mhz=None):
    """The arithmetic floor of the kernel's cell loop: what a SIMD needs for the loop's instruction mix ALONE (static VALU
    count of the code object, profiles/tools/valu_count.py; cost of that mix measured on the GPU at the launch's waves per
    SIMD without loads or cross-lane work, profiles/tools/valu_rate.hip), times the cell iterations one SIMD executes
    (gene blocks x cells / (CUs x 4 SIMDs)), at the clock this process reads."""
    try:
        vm = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))
        ent = vm["kernels"][engine.stats["main_kernel"]]
    except Exception:
        return None
    ncu = torch.cuda.get_device_properties(engine.device).multi_processor_count
    gbw = 64 * ent["genes_per_lane"]
    iters = ((engine.spec.Ng + gbw - 1) // gbw) * engine.Nc_local / (ncu * 4.0)
    waves = max(1, -(-int(engine.stats["main_grid"]) // ncu))          # 256-thread workgroups: one wave per SIMD each
    key = min(ent["floor_ns_per_cell_iter"], key=lambda w: (abs(int(w) - waves), int(w)))
    scale = (vm["mix_clock_ghz"] * 1e3 / clock_mhz) if clock_mhz else 1.0
    bound_us = ent["floor_ns_per_cell_iter"][key] * scale * iters * 1e-3
    return {"bound_us": round(bound_us, 1), "frac": round(bound_us / (kernel_avg_s * 1e6), 4),
            "valu_per_cell_iter": ent["valu_per_cell_iter"], "transcendental_per_cell_iter": ent["trans_per_cell_iter"],
            "waves_per_simd": waves, "floor_ns_per_cell_iter": ent["floor_ns_per_cell_iter"][key],
            "floor_measured_at_mhz": vm["mix_clock_ghz"] * 1e3, "priced_at_mhz": clock_mhz,
            "cell_iters_per_simd": round(iters, 1),
            "source": "profiles/valu_model.json: static instruction count of the cell loop x the cost of that instruction mix "
                      "alone on one SIMD (profiles/tools/valu_rate.hip, no loads, no cross-lane work)"}


def kernel_roofline(engine, run, steps, step_s=None):
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
    clock_mhz = engine.device_clock_mhz()                  # shader clock right after those launches
    st = engine.stats
    achieved = st["algorithmic_bytes"] / avg_s / 1e9
    traffic, traffic_src = None, None
    try:     # PMC numbers cannot be collected from inside the process: they come from the committed rocprofv3 passes, and
        # only when that file was collected on THIS kernel (name) built from THESE sources (csrc hash) at this workload
        tj = json.load(open(os.path.join(ROOT, "profiles", "latest_traffic.json")))
        ent = tj["kernels"].get(st["main_kernel"])
        if (ent and tj.get("workload") == f"{engine.spec.Nc}x{engine.spec.Ng}" and engine.world_size == 1
                and tj.get("csrc_sha16") == csrc_sha16()):
            traffic = int(ent["traffic_bytes"])
            traffic_src = ("profiles/latest_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per gfx950 "
                           f"note; collected on kernel sources {tj['csrc_sha16']} = this build)")
        elif ent:
            traffic_src = ("none: profiles/latest_traffic.json was collected on other kernel sources "
                           f"({tj.get('csrc_sha16')} vs this build {csrc_sha16()}) or another workload")
    except Exception:
        pass
    pipe = (traffic if traffic else st["streamed_bytes"]) / avg_s / 1e9
    vb = valu_bound(engine, avg_s, clock_mhz)
    if vb is not None:
        # the same floor priced at the clock the kernel itself holds inside its cell loop (stamped build,
        # profiles/tools/wave_timeline.py; profiles/valu_model.json "in_loop_clock_mhz"): the light one-wave probe above
        # reads the clock of an idle-ish chip, the likelihood kernel runs 5-10 % below it (DVFS give-back)
        try:
            ilc = json.load(open(os.path.join(ROOT, "profiles", "valu_model.json")))["in_loop_clock_mhz"].get(st["main_kernel"])
        except Exception:
            ilc = None
        if ilc:
            b_us = vb["floor_ns_per_cell_iter"] * (vb["floor_measured_at_mhz"] / ilc) * vb["cell_iters_per_simd"] * 1e-3
            vb["in_loop_clock_mhz"] = ilc
            vb["bound_us_at_in_loop_clock"] = round(b_us, 1)
            vb["frac_at_in_loop_clock"] = round(b_us / (avg_s * 1e6), 4)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # the same algorithmic bytes against the WHOLE driver-timed step (likelihood kernel + the small launches around it):
            # what a user of fit() gets, next to the kernel-level fraction
            "step_frac": None if not step_s else round(st["algorithmic_bytes"] / step_s / 1e9 / HBM_PEAK_GBS, 4),
            "step_overhead_us": None if not step_s else round((step_s - avg_s) * 1e6, 2),
            "traffic": traffic, "traffic_source": traffic_src,
            "hbm_pipe_GBs": round(pipe, 1), "hbm_pipe_frac": round(pipe / HBM_PEAK_GBS, 4),
            "valu": vb,
            "kernel": st["main_kernel"], "kernel_avg_us": round(avg_s * 1e6, 2), "launches": int(n),
            "algorithmic_bytes_per_launch": int(st["algorithmic_bytes"]),
            "streamed_bytes_per_launch": int(st["streamed_bytes"]),
            "count_storage": st["count_storage"],
            "streamed_GBs": round(st["streamed_bytes"] / avg_s / 1e9, 1),
            "note": "achieved / frac price the ALGORITHMIC bytes (the reference's float32 count matrices read once, SURVEY 8d: "
                    "what the task's roofline contract asks for) against the 8 TB/s HBM spec; hbm_pipe_* is what the HBM pipe "
                    "really carries per launch (PMC traffic when a committed rocprofv3 pass matches this workload, else the "
                    "streamed bytes of the uint16 / float32 layout) -- with count_storage u16 that is about half of the "
                    "algorithmic bytes, so frac is an efficiency against the float32 roofline, NOT pipe utilisation; the "
                    "nearer ceiling is the arithmetic floor of the cell loop (valu: the loop's instruction mix alone on one "
                    "SIMD, measured; frac = at the clock this process's probe reads, frac_at_in_loop_clock = at the clock the "
                    "kernel holds inside its loop)",
            "method": "hipEvents recorded by the library on the launch stream around that kernel only, over eager SVI "
                      "steps run right after the timed region"}


def cpu_baseline(args, mode, device=None):
    """Oracle restatement (kind "port": the reference's Pyro path cannot run here) timed on the host cores.
    The op-by-op path streams ~110 full-size (Ng, Nc) temporaries per step, so its cost per cell depends on whether they
    fit the host's last-level cache: it is timed at a 10 000-cell sample AND, when the host has the memory for it
    (>= 2.5 x the ~110 temporaries), at the FULL size, which is then the reported value; otherwise the value is the
    10 000-cell time scaled linearly in cells and says so.  The cells-exponent between the sizes is reported."""
    import math
    from oracle import velocycle_oracle as orc
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    n_mid = min(args.cpu_sample_cells, args.cells)
    n_small = max(n_mid // 4, 1)

    def problem(nsample):
        spec = (make_phase_spec(nsample, args.genes, seed=0, device="cpu") if mode == "phase"
                else make_velocity_spec(nsample // args.conditions, args.genes, mode, args.conditions, 1, seed=0, device="cpu"))
        kw = {}
        for k, v in spec.__dict__.items():
            if k in ("truth", "S_csr", "U_csr"):
                continue
            kw[k] = v.contiguous() if isinstance(v, torch.Tensor) else v
        return spec, orc.Problem(**kw)

    def stepper(p):
        gen = torch.Generator().manual_seed(0)
        first = orc.draw_eps(p, gen)
        st = {"params": orc.init_params(p, first.get("_cov_factor_draw"))}
        opt = orc.ClippedAdam({"lr": 0.03, "lrd": 0.999, "betas": (0.8, 0.99)})

        def one():
            eps = orc.draw_eps(p, gen)
            loss, grads, _, _ = orc.loss_and_grads(p, st["params"], eps)
            if "first" not in st:      # (params, eps, loss, gradients) of the first evaluation: the ELBO-match input at this size
                st["first"] = ({k: v.detach().clone() for k, v in st["params"].items()}, eps, float(loss),
                               {k: v.detach().clone() for k, v in grads.items()})
            st["params"] = opt.step(st["params"], grads)
        return one, st, gen

    spec_s, p_s = problem(n_small)
    # ELBO-match (the other half of BASELINE.json's metric): the HIP path and the port on identical (params, eps)
    one_s, st_s, gen_s = stepper(p_s)
    eps0 = orc.draw_eps(p_s, gen_s)
    loss_cpu, grads_cpu, _, _ = orc.loss_and_grads(p_s, st_s["params"], eps0)
    elbo_match = None

    def hip_eval(spec, params, eps):
        """(-ELBO, {block: gradient}) of the HIP engine (unfused kernel sequence, host eps) on exactly these parameters and draws."""
        from velocycle_amd.engine import HipEngine
        from velocycle_amd.tuning import Tuning
        eng = HipEngine(spec, device=device, tuning=Tuning.from_env())
        eng.set_params({k: v.float() for k, v in params.items()})
        eng.elbo_grad(eps=eng.pack_eps({k: v.float() for k, v in eps.items() if not k.startswith("_")}))
        torch.cuda.synchronize(device)
        out = eng.loss(), {k: v.detach().double().cpu() for k, v in eng.named(eng.grad).items()}
        eng.close()
        return out

    def grad_match(g_hip, g_cpu):
        """Per parameter block: max |HIP - port| / max |port| over the block (both float32 evaluations of the same step), and the
        share of the block's elements that agree to 1e-3 of their OWN magnitude (small elements are invisible to a max-norm)."""
        out, worst = {}, 0.0
        for k, g in g_hip.items():
            w = g_cpu[k].double().reshape(g.shape)
            fin = torch.isfinite(w)
            if not bool(fin.any()) or float(w[fin].abs().max()) == 0.0:
                continue                      # (a block without a path to the loss: conditioned site)
            scale = float(w[fin].abs().max())
            err = (g[fin] - w[fin]).abs()
            out[k] = {"max_err_over_max": float(err.max()) / scale,
                      "share_within_1e-3_elementwise": float((err <= 1e-3 * w[fin].abs() + 1e-6 * scale).double().mean())}
            worst = max(worst, out[k]["max_err_over_max"])
        return out, worst

    if device is not None:
        loss_hip, g_hip = hip_eval(spec_s, st_s["params"], eps0)
        gm, gw = grad_match(g_hip, grads_cpu)
        elbo_match = {"loss_hip": loss_hip, "loss_cpu_port": float(loss_cpu),
                      "rel_err": abs(loss_hip - float(loss_cpu)) / abs(float(loss_cpu)),
                      "grad_rel_err": {k: round(v["max_err_over_max"], 7) for k, v in gm.items()}, "grad_rel_err_max": round(gw, 7),
                      "note": f"one ELBO + gradient evaluation on the {n_small}-cell sample with identical params and eps; the port "
                              "runs in float32, so this bounds both sides' rounding (tests compare against float64 at 1e-5 / 2e-3)"}
    # the op-by-op torch path does not scale to every hardware thread of a big host: pick the thread count that is
    # fastest on this box (short sweep on the mid-size sample, which is past the cache-resident regime), then time with it
    default_nt = torch.get_num_threads()
    del p_s, st_s, one_s
    _, p_m = problem(n_mid)
    one_m, _, _ = stepper(p_m)
    one_m()
    sweep = {}
    for nt in sorted({16, 32, 64, min(default_nt, 128)}):
        if nt > (os.cpu_count() or nt):
            continue
        torch.set_num_threads(nt)
        one_m()
        t0 = time.perf_counter()
        one_m()
        sweep[nt] = time.perf_counter() - t0
    best_nt = min(sweep, key=sweep.get)
    torch.set_num_threads(best_nt)

    def timed(one, budget_s, nmin=3, nmax=40):
        t0 = time.perf_counter()
        n = 0
        while n < nmin or (time.perf_counter() - t0 < budget_s and n < nmax):
            one()
            n += 1
        return n, time.perf_counter() - t0
    nm, dtm = timed(one_m, 4.0)
    t_mid = dtm / nm
    del p_m, one_m
    # full size, if the host can hold it
    nmat = 1 if mode == "phase" else 2
    need = 2.5 * 55 * nmat * 4.0 * args.genes * args.cells
    try:
        import psutil
        avail = float(psutil.virtual_memory().available)
    except Exception:
        avail = 0.0
    full = None
    elbo_match_full = None
    if args.cells > n_mid and avail >= need:
        spec_f, p_f = problem(args.cells)
        one_f, st_f, _ = stepper(p_f)
        one_f()                                           # warm-up (allocator); its loss is kept for the ELBO-match
        nf, dtf = timed(one_f, 0.0, nmin=3, nmax=3)
        full = (nf, dtf)
        if device is not None:
            # the metric's "ELBO-match" at the QUOTED configuration: the port's first full-size evaluation against the HIP
            # engine on the same (params, eps) -- float32 on both sides
            par_f, eps_f, loss_f, grads_f = st_f["first"]
            loss_hip_f, g_hip_f = hip_eval(spec_f, par_f, eps_f)
            gm_f, gw_f = grad_match(g_hip_f, grads_f)
            elbo_match_full = {"cells": args.cells, "genes": args.genes, "loss_hip": loss_hip_f, "loss_cpu_port": loss_f,
                               "rel_err": abs(loss_hip_f - loss_f) / abs(loss_f),
                               # every gradient block of the SAME evaluation, HIP against the port: max |difference| over
                               # the block's max-norm -- the bar of the parity tests is 2e-3 -- and how many elements agree to 1e-3 of
                               # their own magnitude
                               "grad_rel_err": {k: round(v["max_err_over_max"], 7) for k, v in gm_f.items()},
                               "grad_rel_err_max": round(gw_f, 7),
                               "grad_share_within_1e-3_elementwise": {k: round(v["share_within_1e-3_elementwise"], 5) for k, v in gm_f.items()},
                               "grad_bar": 2e-3,
                               "note": "one ELBO + gradient evaluation at the full benchmark size with identical params and eps (the "
                                       "port's warm-up step); float32 on both sides, so each side carries its own rounding through "
                                       "the relu kink of ElogU (profiles/r06_kink_error.md); the float64 comparison at this size is "
                                       "tests/test_hip_fullsize.py"}
        del p_f, one_f, st_f, spec_f
    torch.set_num_threads(default_nt)
    if full is not None:
        t_full = full[1] / full[0]
        expo = math.log(t_full / t_mid) / math.log(args.cells / n_mid)
        how = (f"FULL size {args.cells} cells x {args.genes} genes: {full[0]} steps in {full[1]:.1f}s = {t_full:.2f} s/step (the value); "
               f"{n_mid}-cell sample: {nm} steps in {dtm:.1f}s = {t_mid:.3f} s/step -> measured cost exponent in cells {expo:.2f}")
    else:
        t_full = t_mid * args.cells / n_mid
        expo = None
        how = (f"first {n_mid} of {args.cells} cells x {args.genes} genes: {nm} steps in {dtm:.1f}s = {t_mid:.3f} s/step, scaled "
               f"LINEARLY in cells (the host has {avail / 2 ** 30:.0f} GiB available, the full size needs ~{need / 2 ** 30:.0f} GiB)")
    return {"value": round(1.0 / t_full, 4), "unit": "SVI steps/s", "cores": best_nt,
            "kind": "port", "elbo_match": elbo_match, "elbo_match_full": elbo_match_full, "cells_exponent": None if expo is None else round(expo, 3),
            "sample": "oracle (op-by-op torch fp32 + autograd + ClippedAdam), " + how + f"; {best_nt} torch threads = fastest of "
                      f"sweep { {k: round(v, 3) for k, v in sweep.items()} } s/step on {n_mid} cells; host cpu_count={os.cpu_count()}"}


def device_identity(device):
    pr = torch.cuda.get_device_properties(device)
    return {"name": pr.name, "uuid": str(getattr(pr, "uuid", "")), "gcn_arch": getattr(pr, "gcnArchName", ""),
            "cus": pr.multi_processor_count, "hbm_gb": round(pr.total_memory / 2 ** 30, 1)}


def deadline_s() -> float:
    """N > 1 only: how long a run may take before it is given up (VC_BENCH_DEADLINE_S, default 900 s).  A stuck collective
    or a dead peer otherwise hangs every rank for as long as the caller is willing to wait."""
    return float(os.environ.get("VC_BENCH_DEADLINE_S", "900"))


def arm_watchdog(rank: int):
    import threading

    def fire():
        sys.stderr.write(f"[bench.py rank {rank}] no result after {deadline_s():.0f} s -- a collective or a peer is stuck; exit 124\n")
        sys.stderr.flush()
        os._exit(124)                                  # plain exit of this rank (the launcher then ends the others); no exec

    t = threading.Timer(deadline_s(), fire)
    t.daemon = True
    t.start()
    return t


def self_launch(n: int) -> int:
    import socket
    import subprocess
    with socket.socket() as so:                       # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks carry their own watchdog (arm_watchdog); the parent only waits a little longer than they do
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        so_, se_ = child.communicate(timeout=deadline_s() + 120)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, 9)                       # exactly the process group started above
        so_, se_ = child.communicate()
        se_ += f"\n[bench.py] the {n}-rank job did not finish within {deadline_s() + 120:.0f} s and was ended\n"
    proc = subprocess.CompletedProcess(cmd, child.returncode, so_, se_)
    sys.stderr.write(proc.stderr)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    js = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if ln not in js:
            sys.stderr.write(ln + "\n")               # anything else the ranks printed goes to stderr: stdout is ONE JSON line
    if js:
        print(js[-1])
    return proc.returncode if (proc.returncode != 0 or js) else 1


def main():
    args = parse()
    if args.repeats <= 0:
        args.repeats = max(9, -(-1000 // max(args.steps, 1)))
    # torchrun sets these before this process touches the GPU; nothing below re-executes the process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world and dist_on:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and not dist_on:
        # `python bench.py --gpus N` as written: this process has made NO GPU call yet (importing torch does not initialise
        # HIP), so it may start the launcher as a CHILD process -- one rank per GPU over RCCL -- and relay rank 0's JSON line
        # and the exit code.  Never os.exec*: a process that touched the GPU must not be replaced (Environment notes).
        sys.exit(self_launch(args.gpus))
    # test hook (tests/test_hip_multiproc.py): VC_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and exchanges through
    # gloo, so that the N > 1 path can be exercised end to end on a 1-GPU box.  Never set in a measured run.
    one_device = os.environ.get("VC_BENCH_ONE_DEVICE", "0") == "1"
    device = torch.device("cuda:0" if one_device else f"cuda:{local_rank}")
    torch.cuda.set_device(device)
    # VC_BENCH_NCCL_GROUP=1: a 1-rank nccl group at N = 1, so that the N > 1 code path (all-reduce between the gradient
    # kernels and the optimiser) can be timed on one GPU and compared with the plain N = 1 line (profiles/)
    solo_group = (not dist_on) and os.environ.get("VC_BENCH_NCCL_GROUP", "0") == "1"
    if dist_on:
        arm_watchdog(rank)
    if dist_on or solo_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    from velocycle_amd.engine import HipEngine
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec

    optim = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / 10000), "betas": (0.80, 0.99)}

    def build(mode, cells=None, conditions=None, hw=1, tuning_kw=None):
        cells = args.cells if cells is None else cells
        conditions = args.conditions if conditions is None else conditions
        t0 = time.perf_counter()
        if mode == "phase":
            spec = make_phase_spec(cells, args.genes, seed=0, device=device)
        else:
            spec = make_velocity_spec(cells // conditions, args.genes, mode, conditions, hw, seed=0, device=device)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        tun = Tuning.from_env()
        if tuning_kw:
            tun = tun.replace(**tuning_kw)
        eng = HipEngine(spec, device=device, rank=rank, world_size=world, tuning=tun)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        # N > 1: the fused step cut at its one exchange (K_main -> phase A -> sum over ranks -> phase B).  On the nccl backend
        # the engine owns the exchange (its own RCCL communicator): every launch and every all-reduce of a timed region is
        # enqueued from one C call, no Python and no graph in the loop.  VC_EXCHANGE=torch routes the sum through
        # torch.distributed.all_reduce between the two phases; VC_BENCH_DIST_GRAPH=1 additionally replays that sequence from a
        # hipGraph (opt-in: it has only ever run on a 1-rank RCCL group); VC_ADAM_IMPL_DIST=hip is round 2's five-kernel step.
        graph = None
        if args.no_graph or one_device:
            graph = False
        elif dist_on or solo_group:
            graph = os.environ.get("VC_BENCH_DIST_GRAPH", "0") == "1"
        run = SVIRunner(eng, optim, mode="perf", seed=0, use_graph=graph, force_reduce=solo_group)
        return spec, eng, run, {"synthetic_data_s": round(t1 - t0, 3), "engine_setup_s": round(t2 - t1, 3)}

    spec, eng, run, setup = build(args.mode)
    clock_before = eng.device_clock_mhz()
    times = time_steps(run, args.steps, args.warmup, dist_on, device, args.repeats)
    clock_after = eng.device_clock_mhz()
    dt = median(times)
    sps = args.steps / dt
    losses = run.perf_losses()
    ok, first_bad, n_bad = eng.status()
    n_timed = args.warmup + args.steps * len(times)          # steps run up to the end of the last timed region
    roof = kernel_roofline(eng, run, args.roofline_launches, dt / args.steps)
    if dist_on:
        roof["note"] = f"per-rank kernel on {eng.Nc_local} of {args.cells} cells"
    # "loss read back each step unless stated" -- the headline keeps the losses on the device and reads them
    # once at the end; this is the same step with the loss copied to the host after every step
    t0 = time.perf_counter()
    nrb = 100
    for _ in range(nrb):
        run.run_perf(1, sync=False)
        float(run.loss_hist[run.step_idx - 1].item())
    rb_sps = nrb / (time.perf_counter() - t0)
    # ... and with the loss ring in pinned host memory, written by the device and polled by the host (what fit() does when
    # it has to look at every loss: early_exit / store_output); single-rank fused path
    rb_host = None
    if not dist_on and not solo_group and run.adam_impl == "fused3" and not run.use_graph:
        run.step_with_loss()
        t0 = time.perf_counter()
        ls = [run.step_with_loss() for _ in range(nrb)]
        rb_host = nrb / (time.perf_counter() - t0)
        assert all(l == l for l in ls) or n_bad > 0
    out = {
        "metric": baseline_metric() if args.mode != "phase" else "SVI steps/sec, phase_inference",
        "value": round(sps, 2), "unit": "SVI steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": len(times), "repeat_ms_per_step": [round(1e3 * t / args.steps, 4) for t in times],
        "ms_per_step_min_max": [round(1e3 * min(times) / args.steps, 4), round(1e3 * max(times) / args.steps, 4)],
        "timing": "value / ms_per_step = MEDIAN of `repeats` timed regions of exactly `steps` steps each (barrier + "
                  "synchronize on both sides, max over ranks); repeat_ms_per_step lists all of them in order",
        "device_clock_mhz": {"before_timed_region": round(clock_before, 1), "after_timed_region": round(clock_after, 1),
                             "method": "shader clock: s_memtime ticks per 200 us of s_memrealtime, one wave "
                                       "(vc_device_clock_mhz) -- the clock of a LIGHT kernel; the sustained clock inside the "
                                       "likelihood kernel is lower (2.29-2.36 GHz at steady state, 1.6-1.7 GHz during the first "
                                       "~30 ms of a fresh process: profiles/r02_wave_timeline_u16.txt), and that ramp shows in "
                                       "the first entries of repeat_ms_per_step"},
        "setup_s": setup,
        "config": {"workload": f"synthetic {args.cells} cells x {args.genes} genes " + ("phase_inference" if args.mode == "phase" else "velocity_inference") + ", "
                               + {"vjoint": "mean-field guide, nothing conditioned (every gradient)",
                                  "vcond": "tutorial flow: LRMN guide conditioned on phixy, nu, shape_inv",
                                  "vcond_mf": "mean-field guide conditioned on phixy, nu, shape_inv",
                                  "phase": "spliced matrix only, mean-field guide, nothing conditioned"}[args.mode]
                               + ", NegativeBinomial noise, H=1, Hw=1"
                               + (f", {args.conditions} samples (Nx = Nb = {args.conditions}, per-batch offsets)" if args.conditions > 1 and args.mode != "phase" else ""),
                   "cells": args.cells, "genes": args.genes, "mode": args.mode, "conditions": args.conditions,
                   "launches_per_step": eng.stats.get("launches_per_step"),
                   "small_kernels": eng.stats.get("tail_spec_name"),
                   "nu_omega_partials_from_main_kernel": bool(eng.stats.get("pw_inline")),
                   "parallelism": f"cells sharded over {world} GPU(s), one all-reduce of gene-level gradients per step",
                   "step": ("Philox eps -> ELBO+grad (HIP kernels) -> " + (("gloo (test hook) " if one_device else "RCCL ") + "all-reduce"
                                                                            + (f" [{run.exchange}]" if run.exchange else "") + " -> " if (dist_on or solo_group) else "")
                            + f"ClippedAdam ({run.adam_impl}), " + ("hipGraph replay" if run.use_graph else "eager launches")
                            + "; losses stay in a device ring and are read back ONCE after the timed region "
                              "(with_loss_readback_each_step gives the rate with a host read-back after every step)")},
        "with_loss_readback_each_step": {"value": round(rb_host if rb_host else rb_sps, 2), "unit": "SVI steps/s", "steps": nrb,
                                         "how": ("loss ring in pinned host memory, written by the device, polled by the host "
                                                 "(SVIRunner.step_with_loss)" if rb_host else "stream synchronise + copy of the loss"),
                                         "by_sync_and_copy": round(rb_sps, 2)},
        "roofline": roof,
        "loss_first_last": [losses[0], losses[n_timed - 1]],
        "nonfinite_loss_steps": n_bad,
    }
    if dist_on or solo_group:
        import torch.distributed as dist
        me = dict(device_identity(device), rank=rank, local_rank=local_rank, cells=eng.Nc_local, pid=os.getpid())
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, me)
        out["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ids,
                              "distinct_devices": len({(i["uuid"], i["local_rank"]) for i in ids}),
                              "step_launch": "hipGraph replay (RCCL all-reduce captured)" if run.use_graph else "eager",
                              "step_kind": run.adam_impl, "exchange": run.exchange,
                              "exchange_check": getattr(run, "exchange_check", None)}
    else:
        out["device"] = device_identity(device)
    if dist_on and not args.no_weak:
        # Weak scaling from the same invocation: --cells cells PER RANK (N x the strong problem), the same
        # kernels and the same exchange; the strong run above stays the headline (the north-star target is phrased on the fixed
        # 50k x 2k problem), this block says what a rank's step costs when its shard does not shrink with N.
        del run, eng, spec
        torch.cuda.empty_cache()
        spec_w, eng_w, run_w, setup_w = build(args.mode, cells=args.cells * world)
        tw = time_steps(run_w, args.steps, args.warmup, dist_on, device, max(5, args.repeats))
        rf_w = kernel_roofline(eng_w, run_w, args.roofline_launches, median(tw) / args.steps)
        out["weak"] = {"scaling": "weak", "cells_per_rank": eng_w.Nc_local, "cells_total": args.cells * world, "genes": args.genes,
                       "value": round(args.steps / median(tw), 2), "unit": "SVI steps/s", "ms_per_step": round(1e3 * median(tw) / args.steps, 4),
                       "repeat_ms_per_step": [round(1e3 * t / args.steps, 4) for t in tw],
                       "cells_per_s": round(args.cells * world * args.steps / median(tw), 1),
                       "kernel": rf_w["kernel"], "kernel_avg_us": rf_w["kernel_avg_us"], "frac": rf_w["frac"],
                       "step_frac": rf_w["step_frac"], "exchange": run_w.exchange, "setup_s": setup_w,
                       "note": "every rank keeps --cells cells (the strong problem's size per GPU); value = steps/s of the N-times "
                               "larger job; compare ms_per_step with the 1-GPU strong line"}
        del run_w, eng_w, spec_w
        torch.cuda.empty_cache()
        run = eng = spec = None
    def loss_every_demo(engine):
        # OPT-IN, never the headline and not what the reference does (it reads the loss of every step): the loss formed at every
        # 10th step only -- the other steps run the gradient-only instantiation of the likelihood kernel (vc_set_loss_every;
        # DESIGN.md section 8).  Reported under its own name next to the default numbers of the same model.
        try:
            r3 = SVIRunner(engine, optim, mode="perf", seed=0, loss_every=10)
            ts3 = time_steps(r3, args.steps, args.warmup, False, device, args.repeats)
            rf3 = kernel_roofline(engine, r3, args.roofline_launches, median(ts3) / args.steps)
            res = {"steps_per_s": round(args.steps / median(ts3), 2), "ms_per_step": round(1e3 * median(ts3) / args.steps, 4),
                   "kernel_avg_us_mix": rf3["kernel_avg_us"], "step_overhead_us": rf3["step_overhead_us"],
                   "note": "losses hold NaN at 9 of 10 steps; 9 of 10 likelihood launches are the gradient-only instantiation "
                           "(kernel_avg_us_mix averages both)"}
            SVIRunner(engine, optim, mode="perf", seed=0, init=False)      # (restores the engine's default: every loss)
            return res
        except Exception as ex:            # (a configuration without such a kernel: say so, do not fail the line)
            return {"error": str(ex)[:200]}

    extra = {}
    if not dist_on and not solo_group and not args.no_loss_every_demo and rank == 0:
        out["opt_in_loss_every_10"] = loss_every_demo(eng)
    if not args.no_extra_modes and not dist_on:
        del run, eng, spec
        torch.cuda.empty_cache()
        for m in [x for x in ("vcond", "phase", "vjoint") if x != args.mode][:2]:
            s2, e2, r2, _ = build(m)
            ts2 = time_steps(r2, args.steps, args.warmup, False, device, args.repeats)
            rf = kernel_roofline(e2, r2, args.roofline_launches, median(ts2) / args.steps)
            extra[m] = {"steps_per_s": round(args.steps / median(ts2), 2), "ms_per_step": round(1e3 * median(ts2) / args.steps, 4),
                        "launches_per_step": e2.stats.get("launches_per_step"), "small_kernels": e2.stats.get("tail_spec_name"),
                        "kernel": rf["kernel"],
                        "kernel_avg_us": rf["kernel_avg_us"], "hbm_achieved_GBs": rf["achieved"],
                        "hbm_frac": rf["frac"], "step_frac": rf["step_frac"], "step_overhead_us": rf["step_overhead_us"], "hbm_pipe_frac": rf["hbm_pipe_frac"], "valu_frac": (rf["valu"] or {}).get("frac"),
                        "valu_frac_at_in_loop_clock": (rf["valu"] or {}).get("frac_at_in_loop_clock")}
            if not args.no_loss_every_demo:
                extra[m]["opt_in_loss_every_10"] = loss_every_demo(e2)
            del s2, e2, r2
            torch.cuda.empty_cache()
        # BASELINE configs[4]: two samples (Nx = Nb = 2, one angular speed and one batch offset per sample) at the same total size --
        # the one-hot batch design is folded per workgroup of the likelihood kernel (NB = 0 instantiation: nothing per cell)
        # ... the tutorials' FIRST velocity stage: constant angular speed (omega_n_harmonics = 0, AngularSpeed.trivial_prior(harmonics=0):
        # Tutorial_Capolupo_HumanFibroblasts_OneSample.ipynb:690,721), one and two samples -- the most-run workload of the reference
        # ... and the headline model with the counts STORED as float32 (Tuning(count_storage="f32")): the kernel then really moves the
        # float32 bytes the roofline prices (narrower storage "must be reported separately")
        for name, m, nc, hw, tk in (("vjoint_2sample", "vjoint", 2, 1, None), ("vcond_2sample", "vcond", 2, 1, None),
                                    ("vcond_hw0", "vcond", 1, 0, None), ("vcond_hw0_2sample", "vcond", 2, 0, None),
                                    ("vjoint_f32_storage", "vjoint", 1, 1, {"count_storage": "f32"})):
            s2, e2, r2, _ = build(m, conditions=nc, hw=hw, tuning_kw=tk)
            # (the same number of timed regions as the headline: with fewer, the median falls into the ~30 ms the shader clock needs to
            # ramp after every engine setup, and the mode reads 3-5 us per step slow -- round 6, profiles/r06_bench_protocol.md)
            ts2 = time_steps(r2, args.steps, args.warmup, False, device, args.repeats)
            rf = kernel_roofline(e2, r2, args.roofline_launches, median(ts2) / args.steps)
            extra[name] = {"steps_per_s": round(args.steps / median(ts2), 2), "ms_per_step": round(1e3 * median(ts2) / args.steps, 4),
                           "launches_per_step": e2.stats.get("launches_per_step"), "kernel": rf["kernel"],
                           "onehot_batches": e2.stats.get("onehot_batches"), "small_kernels": e2.stats.get("tail_spec_name"),
                           "omega_harmonics": hw, "count_storage": e2.stats.get("count_storage"),
                           "kernel_avg_us": rf["kernel_avg_us"], "hbm_achieved_GBs": rf["achieved"], "hbm_frac": rf["frac"],
                           "step_frac": rf["step_frac"], "step_overhead_us": rf["step_overhead_us"],
                           "streamed_GBs": rf["streamed_GBs"],
                           "hbm_pipe_frac": rf["hbm_pipe_frac"], "valu_frac": (rf["valu"] or {}).get("frac")}
            del s2, e2, r2
            torch.cuda.empty_cache()
        out["modes"] = extra
    if rank == 0 and not dist_on and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.mode, device)
        out["elbo_match_full"] = out["cpu_baseline"].get("elbo_match_full")     # the metric's second half, at the quoted size
    if dist_on or solo_group:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out, ensure_ascii=False))


if __name__ == "__main__":
    main()
