"""SVI driver on top of `HipEngine`: the loop body of `svi.step` (reference
phase_inference_model.py:169 / velocity_inference_model.py:120), i.e.
ELBO+gradient (HIP) -> one all-reduce of the replicated-parameter gradients when cells are sharded
over GPUs (RCCL through torch.distributed) -> ClippedAdam on ONE flat parameter tensor (PyTorch ops).

`ClippedAdam` here restates pyro.optim.ClippedAdam (pyro-ppl 1.8.6, optim/clipped_adam.py) for a flat
buffer: lr <- lr*lrd before every update, elementwise clamp of the gradient to +-clip_norm,
m/v moments, step lr*sqrt(1-b2^t)/(1-b1^t), denominator sqrt(v)+eps (eps outside the sqrt, unlike
torch.optim.Adam's bias-corrected form).  Pyro keeps one optimiser per parameter tensor, all created
at step 0 with identical hyper-parameters, so a single flat state is equivalent.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .engine import HipEngine
from .rng import draw_eps


class ClippedAdam:
    """Drop-in for `pyro.optim.ClippedAdam({...})` as the tutorials build it; only a holder of the
    argument dict (`pt_optim_args`, same attribute name as PyroOptim)."""

    def __init__(self, optim_args: dict, clip_args=None):
        if clip_args:
            raise NotImplementedError("PyroOptim clip_args (gradient clipping by norm / value in front of the optimiser) are not supported")
        self.pt_optim_args = dict(optim_args)


class Adam(ClippedAdam):
    """Drop-in for `pyro.optim.Adam({...})` (= torch.optim.Adam; tutorials/1D_Pancreas_Analysis.ipynb cell 26)."""


_CLIPPED_ADAM_KEYS = {"lr", "betas", "eps", "weight_decay", "clip_norm", "lrd"}
_ADAM_KEYS = {"lr", "betas", "eps", "weight_decay", "amsgrad", "maximize", "foreach", "capturable", "differentiable", "fused"}


def optimizer_kind_of(optimizer) -> str:
    """"clipped_adam" | "adam": what kind of optimiser the object handed to fit() is (ADVICE / VERDICT r4: a pyro.optim.Adam used
    to be run as ClippedAdam without a word).  A dict keeps meaning ClippedAdam's argument dict (the tutorials' `{'lr', 'lrd',
    'betas'}`); an object is recognised by the torch optimiser PyroOptim wraps (`pt_optim_constructor`) or by its class; anything
    else is a TypeError that names it -- the engine implements these two optimisers, nothing is substituted silently."""
    if isinstance(optimizer, dict):
        return optimizer.get("_kind", "clipped_adam")
    ctor = getattr(optimizer, "pt_optim_constructor", None)
    name = getattr(ctor, "__name__", None) or type(optimizer).__name__
    if name == "ClippedAdam":
        return "clipped_adam"
    if name == "Adam":
        return "adam"
    raise TypeError(f"optimizer {name!r} is not supported: the HIP engine implements pyro.optim.ClippedAdam and pyro.optim.Adam "
                    "(pass one of them, velocycle_amd.optim.ClippedAdam / Adam, or ClippedAdam's argument dict)")


def optim_args_of(optimizer) -> dict:
    """The validated argument dict of a supported optimiser, with its kind under the key "_kind"."""
    kind = optimizer_kind_of(optimizer)
    if isinstance(optimizer, dict):
        a = dict(optimizer)
    else:
        if getattr(optimizer, "pt_clip_args", None):
            raise NotImplementedError("PyroOptim clip_args are not supported by the HIP engine")
        if not hasattr(optimizer, "pt_optim_args"):
            raise TypeError("optimizer must expose `pt_optim_args` (a PyroOptim) or be a dict")
        a = optimizer.pt_optim_args
        if callable(a):
            raise TypeError("callable optim args (per-parameter arguments) are not supported")
        a = dict(a)
    a.pop("_kind", None)
    allowed = _CLIPPED_ADAM_KEYS if kind == "clipped_adam" else _ADAM_KEYS
    unknown = sorted(set(a) - allowed)
    if unknown:
        raise TypeError(f"{kind}: unknown optimiser argument(s) {unknown} (accepted: {sorted(allowed)})")
    if kind == "adam":
        for k in ("amsgrad", "maximize"):
            if a.get(k):
                raise NotImplementedError(f"torch.optim.Adam({k}=True) is not supported by the HIP engine")
    if float(a.get("weight_decay", 0.0)) < 0.0:
        raise ValueError("weight_decay must be >= 0")
    a["_kind"] = kind
    return a


def frozen_param_names(spec) -> set:
    """Names of the guide's parameter TENSORS that have no path to the loss because every sample site they feed is conditioned
    (`poutine.condition` on the model + `poutine.block(guide, hide=...)`: velocity_inference_model.py:65-66).  Their `.grad` stays
    None in Pyro, PyroOptim skips them, so weight decay must not move them either (without weight decay they stay put by
    themselves: zero gradient, zero moments).  Tensor granularity, as autograd sees it: tests/test_pyro_boundary_cpu.py holds this
    table against the oracle's autograd connectivity on every fixture."""
    c = set(spec.condition_on)
    out = set()
    if "ν" in c:
        out |= {"ν_locs", "ν_scales"}
    if "Δν" in c:
        out.add("Δν_locs")
    if "ϕxy" in c:
        out.add("ϕxy_locs")
    if "shape_inv" in c:
        out.add("shape_inv_locs")
    if spec.kind == "velocity":
        if "logβg" in c:
            out |= {"logβg_locs", "logβg_scales"}
        if spec.guide == "lrmn":
            # X = loc + W eps_W + sqrt(D) eps_D feeds log gamma (X[:Ng]), nu_omega (X[Ng:]) and the conditional mean of log beta
            if {"logγg", "νω", "logβg"} <= c:
                out |= {"loc", "cov_factor", "cov_diag"}
            if {"logβg", "rho_real"} <= c:
                out.add("rho_real_loc")
        else:
            if "logγg" in c:
                out |= {"logγg_locs", "logγg_scales"}
            if "νω" in c:
                out |= {"νω_locs", "νω_scales"}
    return out


class FlatClippedAdam:
    """The optimiser on ONE flat tensor: pyro's ClippedAdam (kind "clipped_adam", the default) or torch's Adam (kind "adam",
    what pyro.optim.Adam wraps), both with optional weight decay (pyro: `grad.add(p, alpha=wd)` behind the clamp; torch: in
    front of the moments).  impl="torch": PyTorch ops on the flat tensor; impl="hip": the library's one-launch kernel
    (vc_adam_update), step counter read from the engine's device counter.  Same arithmetic."""

    def __init__(self, n: int, optim_args: dict, device, capturable: bool = False, impl: str = "torch",
                 engine=None):
        a = dict(optim_args)
        self.kind = a.pop("_kind", "clipped_adam")
        self.lr0 = float(a.get("lr", 1e-3))
        self.b1, self.b2 = (float(x) for x in a.get("betas", (0.9, 0.999)))
        self.eps = float(a.get("eps", 1e-8))
        # torch's Adam neither clamps nor decays: expressed as an infinite clamp and lrd = 1 (what the kernels are handed too)
        self.clip = float(a.get("clip_norm", 10.0)) if self.kind == "clipped_adam" else math.inf
        self.lrd = float(a.get("lrd", 1.0)) if self.kind == "clipped_adam" else 1.0
        self.wd = float(a.get("weight_decay", 0.0))
        # weight decay skips the parameter tensors PyroOptim never steps (frozen_param_names): a float mask over the n updated
        # floats for the PyTorch-op path, a byte mask over the whole flat buffer for the kernels (engine.frozen_mask)
        self.decay_mask = None
        if self.wd != 0.0 and engine is not None and hasattr(engine, "frozen_mask"):
            fm = engine.frozen_mask()
            if fm is not None:
                self.decay_mask = (fm[engine.header:] == 0).to(device)      # True: the entry takes weight decay
        if self.kind == "adam" and capturable and impl == "torch":
            raise ValueError("pyro.optim.Adam with hipGraph replay of the PyTorch-op optimiser is not supported (use the default launches)")
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        self.t = 0
        self.t_vec = None
        self.capturable = capturable
        self.impl, self.engine = impl, engine
        if capturable and impl == "torch":     # step counter and schedule live on the device so the update can be replayed
            self.t_dev = torch.zeros((), dtype=torch.float64, device=device)
            self._c = {k: torch.tensor(v, dtype=torch.float64, device=device)
                       for k, v in dict(lr0=self.lr0, lrd=self.lrd, b1=self.b1, b2=self.b2).items()}

    def set_step_counts(self, t_vec: torch.Tensor):
        """Per-element step counts (a 'mixed' continued fit: pyro_compat).  Refused here, before any step has touched the
        moments, where no implementation applies them: only the PyTorch-op ClippedAdam does."""
        if self.kind != "clipped_adam" or self.impl != "torch" or self.capturable:
            raise RuntimeError("per-parameter step counts (a 'mixed' continued fit) run on the PyTorch-op ClippedAdam only "
                               f"(kind={self.kind!r}, impl={self.impl!r}, capturable={self.capturable})")
        self.t_vec = t_vec

    def step(self, p: torch.Tensor, g: torch.Tensor, t_dev: Optional[torch.Tensor] = None, loss_hdr=None,
             loss_ring=None):
        if self.t_vec is not None and (self.kind != "clipped_adam" or self.impl != "torch" or self.capturable):
            # (a state dict loaded into another kind of optimiser: still refused before m / v move)
            raise RuntimeError("per-parameter step counts (a 'mixed' continued fit) run on the PyTorch-op ClippedAdam only "
                               f"(kind={self.kind!r}, impl={self.impl!r}, capturable={self.capturable})")
        if self.impl == "hip":
            self.t += 1
            fm = self.engine.frozen_mask() if self.wd != 0.0 else None
            self.engine.adam_update(self.kind, p, g, self.m, self.v, self.lr0, self.lrd, self.b1, self.b2, self.eps,
                                    self.clip, self.wd, frozen=None if fm is None else fm[self.engine.header:],
                                    t=self.t, t_dev=t_dev, loss_hdr=loss_hdr, loss_ring=loss_ring)
            return
        if self.kind == "clipped_adam":
            g = g.clamp(-self.clip, self.clip)
        if self.wd != 0.0:
            # only the FROZEN entries are masked: a live parameter sitting at -inf (LRMN cov_factor zeros, stored as logs) decays
            # to NaN here exactly as in vc_adam_elem and in Pyro itself (and trips the non-finite latch) -- ADVICE r5
            g = g.add(p if self.decay_mask is None else torch.where(self.decay_mask, p, torch.zeros((), dtype=p.dtype, device=p.device)),
                      alpha=self.wd)
        self.m.lerp_(g, 1.0 - self.b1)
        self.v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
        if self.kind == "adam":
            # torch.optim.Adam: step lr / (1 - b1^t), denominator sqrt(v) / sqrt(1 - b2^t) + eps
            self.t += 1
            bc1, bc2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
            denom = (self.v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m, denom, value=-self.lr0 / bc1)
            return
        denom = self.v.sqrt().add_(self.eps)
        if self.t_vec is not None:
            # per-element step counts (a fit that continues SOME parameters of an earlier one with the same optimizer object,
            # pyro_compat: PyroOptim keeps one step count per parameter tensor): the schedule of each element is its own
            self.t += 1
            self.t_vec += 1.0
            tv = self.t_vec
            step = self.lr0 * torch.pow(torch.tensor(self.lrd, dtype=torch.float64, device=tv.device), tv) * \
                torch.sqrt(1.0 - torch.pow(torch.tensor(self.b2, dtype=torch.float64, device=tv.device), tv)) / \
                (1.0 - torch.pow(torch.tensor(self.b1, dtype=torch.float64, device=tv.device), tv))
            p.sub_((self.m / denom) * step.float())
            return
        if not self.capturable:
            self.t += 1
            lr = self.lr0 * self.lrd ** self.t
            step_size = lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
            p.addcdiv_(self.m, denom, value=-step_size)
        else:
            self.t_dev += 1.0
            c = self._c
            step_size = c["lr0"] * torch.pow(c["lrd"], self.t_dev) * \
                torch.sqrt(1.0 - torch.pow(c["b2"], self.t_dev)) / (1.0 - torch.pow(c["b1"], self.t_dev))
            p.sub_((self.m / denom) * step_size.float())

    def steps_done(self, step_dev: Optional[torch.Tensor] = None) -> int:
        """1-based Adam step of the last update: the host counter, the torch-op device counter, or (impl="hip" in a
        replayed graph) the engine's device step counter that the kernels read."""
        if self.impl == "torch" and self.capturable:
            return int(self.t_dev.item())
        if self.impl == "hip" and step_dev is not None:
            return int(step_dev.item())
        return self.t

    def state_dict(self, step_dev: Optional[torch.Tensor] = None):
        sd = dict(m=self.m.detach().cpu().clone(), v=self.v.detach().cpu().clone(), t=self.steps_done(step_dev))
        if self.t_vec is not None:      # per-element step counts of a 'mixed' continued fit
            sd["t_vec"] = self.t_vec.detach().cpu().clone()
        return sd

    def load_state_dict(self, sd):
        self.m.copy_(torch.as_tensor(sd["m"]).to(self.m.device))
        self.v.copy_(torch.as_tensor(sd["v"]).to(self.v.device))
        self.t = int(sd["t"])
        self.t_vec = torch.as_tensor(sd["t_vec"]).to(device=self.m.device, dtype=torch.float64).clone() if sd.get("t_vec") is not None else None
        if self.impl == "torch" and self.capturable:
            self.t_dev.fill_(float(self.t))


class SVIRunner:
    """mode="parity": eps drawn on the host in the reference's RNG order (seed-for-seed comparable
    with the reference), loss read back every step.  mode="perf": eps from the in-kernel Philox
    stream, every launch of a run enqueued from one C call (single rank: vc_svi_run_fused; cells sharded:
    vc_svi_run_sharded with the exchange named by `exchange`), losses kept on the device."""

    def __init__(self, engine: HipEngine, optim_args: dict, mode: str = "parity", seed: Optional[int] = None,
                 process_group=None, use_graph: Optional[bool] = None, warmup_draw: bool = True,
                 init: bool = True, adam_impl: Optional[str] = None, force_reduce: bool = False,
                 exchange: Optional[str] = None, num_particles: int = 1, loss_every: int = 1):
        assert mode in ("parity", "perf")
        # loss_every = k > 1 (opt-in, SURVEY.md section 5 "or every k steps in perf mode"): only every k-th step of a fused
        # single-rank run forms the loss, the others run the gradient-only likelihood kernel (engine.set_loss_every: NB noise,
        # fast kernel set; worth + 23 % for the tutorial flow's velocity stage, + 4-7 % where shape_inv is learned);
        # perf_losses() reports NaN for the steps in between.  The reference reads the
        # loss of EVERY step (velocity_inference_model.py:118-121): that is k = 1, the default.
        self.loss_every = int(loss_every)
        if self.loss_every < 1:
            raise ValueError("loss_every must be >= 1")
        # Trace_ELBO(num_particles=K): K guide draws per step, loss and gradients averaged before the optimiser step.  K > 1
        # runs the unfused kernel sequence once per particle (the fused steps draw the NEXT step's single sample inside
        # the optimiser kernel); parity mode draws K host eps sets per step, perf mode uses the Philox streams
        # (seed, step * K + k)
        self.K = int(num_particles)
        if self.K < 1:
            raise ValueError("num_particles must be >= 1")
        self.e, self.mode = engine, mode
        self.pg = process_group
        self.world = engine.world_size
        self.seed = 0 if seed is None else int(seed)
        # force_reduce: issue the all-reduce even with one rank (exercises the RCCL-in-graph path on one GPU)
        self.do_reduce = self.world > 1 or force_reduce
        # perf mode launches plain kernels by default: a single rank enqueues every launch of a run from one C call
        # (vc_svi_run_fused; measured ~6 us per step faster than graph replay, profiles/r02_step_overhead.md), and so do N > 1
        # ranks when the engine owns the exchange (vc_svi_run_sharded, VC_PHASE_AB).  hipGraph replay of the N > 1 step with
        # a torch.distributed collective inside is OPT-IN (use_graph=True): it has only ever run on a 1-rank RCCL group, and
        # the ranks agree on the outcome of the capture (MIN all-reduce) before anyone replays.
        self.use_graph = False if use_graph is None else bool(use_graph)
        if self.do_reduce and self.use_graph:
            import torch.distributed as dist
            if dist.get_backend(process_group) != "nccl":
                # only RCCL collectives are stream work that a hipGraph can capture; gloo (CPU / one-device tests)
                # completes its all-reduce on the host -> eager launches
                self.use_graph = False
        # "torch": PyTorch ops; "hip": one kernel after the gradient; "fused": merged with the last gradient kernel
        # (4 launches per step); "fused3": the three-launch step of vc_svi_step_fused (reductions + chain rule + optimiser +
        # the NEXT step's guide sample in one kernel) -- the single-rank default; "sharded": the same step cut at its one
        # exchange (K_main -> phase A -> sum over ranks -> phase B: vc_svi_run_sharded) -- the default when cells are sharded
        # (engine.tuning.adam_impl / adam_impl_dist override the perf defaults for A/B measurements, e.g. the unfused "hip" sequence)
        tun = getattr(engine, "tuning", None)
        if tun is None:
            from .tuning import Tuning
            tun = Tuning()
        self.tuning = tun
        if self.world > 1:
            # FIRST collective of the runner, in front of every check that may raise on a subset of the ranks (unknown exchange,
            # RCCL / p2p initialisation): a rank that raised later would leave the others waiting in this all-reduce
            self._assert_same_tuning_on_every_rank()
        if adam_impl is None and self.K > 1 and mode == "perf":
            adam_impl = "hip"
        if adam_impl is None and mode == "perf" and (getattr(engine, "stats", None) or {}).get("generic"):
            # a configuration outside the compiled fast set (H > 3, > 4 batches, LRMN rank > 8, > 64 angular-speed
            # coefficients): the run-time-sized kernel set has the unfused sequence only (any number of ranks)
            adam_impl = "hip"
        if self.K > 1 and adam_impl in ("fused", "fused3", "sharded"):
            raise ValueError(f"adam_impl={adam_impl!r} draws one sample per step; num_particles > 1 runs the unfused sequence")
        if adam_impl is None:
            if mode != "perf":
                adam_impl = "torch"
            elif self.do_reduce:
                adam_impl = tun.adam_impl_dist or "sharded"
            else:
                adam_impl = tun.adam_impl or "fused3"
        self.adam_impl = adam_impl
        if self.adam_impl in ("fused", "fused3") and (self.do_reduce or mode != "perf"):
            raise ValueError(f"adam_impl={self.adam_impl!r} needs mode='perf' on a single rank")
        if self.adam_impl == "sharded" and mode != "perf":
            raise ValueError("adam_impl='sharded' needs mode='perf'")
        # who sums the exchange buffer of the sharded step: "engine" = the library's own RCCL communicator, whole runs enqueued
        # from one C call (needs the nccl backend and > 1 rank); "torch" = torch.distributed.all_reduce between the two phases
        # of every step (any backend: gloo in the one-device tests); "p2p" = the one-shot exchange over peer-mapped device
        # memory, also enqueued from one C call (opt-in: VC_EXCHANGE=p2p)
        self.exchange = None
        self.xbuf = None
        if self.adam_impl == "sharded":
            import torch.distributed as dist
            want = exchange or tun.exchange
            if want is None:
                if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
                    want = "none"              # one rank, no process group: nothing to sum, nobody to call all_reduce on
                else:
                    want = "engine" if (self.world > 1 and dist.get_backend(process_group) == "nccl" and not self.use_graph) else "torch"
            if want == "engine":
                if not (self.do_reduce and dist.get_backend(process_group) == "nccl"):
                    raise ValueError("exchange='engine' needs a process group on the nccl (RCCL) backend")
                if not engine.init_rccl_comm(process_group):
                    import warnings
                    warnings.warn("the engine's RCCL communicator could not be created on every rank; the exchange goes "
                                  "through torch.distributed.all_reduce instead")
                    want = "torch"
            elif want == "p2p":                # one-shot exchange over peer-mapped device memory (opt-in, vc_p2p_exchange.hip)
                if not self.do_reduce:
                    raise ValueError("exchange='p2p' needs a process group")
                if not engine.init_p2p_exchange(process_group):
                    raise RuntimeError("the peer-to-peer exchange could not be connected on every rank: "
                                       + engine.lib.vc_last_error(engine._h).decode())
            elif want == "none":               # measurement aid (profiles/tools/step_time_vs_shard.py): one rank, nothing to sum
                if self.world > 1:
                    raise ValueError("exchange='none' is a single-rank measurement aid")
            elif want != "torch":
                raise ValueError(f"unknown exchange {want!r}")
            self.exchange = want
            self.xbuf = torch.zeros(engine.exchange_size(), dtype=torch.float32, device=engine.device)
        optim_args = optim_args_of(optim_args)          # (validated; a plain dict means ClippedAdam's arguments)
        self.opt = FlatClippedAdam(engine.total - engine.header, optim_args, engine.device,
                                   capturable=self.use_graph,
                                   impl=("hip" if self.adam_impl in ("fused", "fused3", "sharded") else self.adam_impl), engine=engine)
        if hasattr(engine, "set_optimizer"):             # what the engine's own step entry points apply (the engine outlives its runners)
            engine.set_optimizer(self.opt.kind, self.opt.wd, engine.frozen_mask() if self.opt.wd != 0.0 else None)
        self._primed = False          # fused3: the tables of the current step have been sampled from the current params
        self.step_idx = 0
        self.losses: List[float] = []
        self._graph = None
        self.gen = None
        if mode == "parity":
            self.gen = torch.Generator().manual_seed(self.seed) if seed is not None else None
        if init:
            cov = None
            if mode == "parity":
                self._first = draw_eps(engine.spec, self.gen) if warmup_draw else None
                self._pending = None
                if engine.spec.kind == "velocity" and engine.spec.guide == "lrmn":
                    if self._first is None:
                        self._pending = draw_eps(engine.spec, self.gen)
                    cov = (self._first or self._pending)["_cov_factor_draw"]
            elif engine.spec.kind == "velocity" and engine.spec.guide == "lrmn":
                g = torch.Generator().manual_seed(self.seed)
                M = engine.spec.Ng + engine.spec.Nx * engine.spec.Nhw
                cov = torch.normal(torch.zeros((M, engine.spec.rho_rank)),
                                   torch.ones((M, engine.spec.rho_rank)) * 0.02, generator=g)
            engine.init_params(cov)
        else:
            self._first, self._pending = None, None
        if mode == "perf":
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=engine.device)
            self.loss_hist = None
        self._arm_loss_every()

    def _assert_same_tuning_on_every_rank(self):
        """Every rank of a sharded run must hold the same Tuning and have derived the same layout from it (one rank with another
        genes_per_lane would pad the genes differently: another exchange buffer, a hang or garbage in the first all-reduce):
        a MIN and a MAX all-reduce of a digest of (tuning, gradient layout, exchange size, kernel), compared on every rank."""
        import hashlib
        import torch.distributed as dist
        e = self.e
        stats = getattr(e, "stats", None) or {}
        try:
            xsize = e.exchange_size()
        except Exception:
            xsize = -1
        # the whole kernel name (one-hot batch detection changes its NB argument, not only genes per lane) and the batch count
        desc = repr((self.tuning.digest(), e.header, e.n_global, stats.get("main_kernel", ""), stats.get("onehot_batches", 0), xsize))
        h = int.from_bytes(hashlib.sha256(desc.encode()).digest()[:8], "little") >> 2
        dev = e.device if dist.get_backend(self.pg) == "nccl" else "cpu"
        lo = torch.tensor([h], dtype=torch.int64, device=dev)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if int(lo.item()) != int(hi.item()):
            from .engine import HipEngineError
            raise HipEngineError(f"rank {e.rank}: the ranks of this sharded run do not share one Tuning / layout "
                                 f"(this rank: {self.tuning}, kernel {stats.get('main_kernel')}, layout {desc})")

    # ------------------------------------------------------------------------------------------
    def _reduce(self):
        if self.do_reduce:
            import torch.distributed as dist
            dist.all_reduce(self.e.grad[: self.e.header + self.e.n_global], group=self.pg)

    def _update(self, t_dev=None, loss_hdr=None, loss_ring=None):
        h = self.e.header
        self.opt.step(self.e.params[h:], self.e.grad[h:], t_dev=t_dev, loss_hdr=loss_hdr, loss_ring=loss_ring)

    def step(self, eps: Optional[Dict[str, torch.Tensor]] = None) -> float:
        """One SVI step in parity mode; returns the loss like `svi.step` does."""
        e = self.e
        if self.K > 1:
            return self._step_particles(eps)
        if eps is None:
            if getattr(self, "_pending", None) is not None:
                eps, self._pending = self._pending, None
            else:
                eps = draw_eps(e.spec, self.gen)
        e.elbo_grad(eps=e.pack_eps(eps), step=self.step_idx)
        self._reduce()
        hdr = e.grad[:2].double().cpu()
        loss = float(e.loss_dev.item()) if self.world == 1 else float(hdr.sum())
        self._update()
        self.step_idx += 1
        self.losses.append(loss)
        return loss

    def _step_particles(self, eps_list=None) -> float:
        """One parity-mode step of Trace_ELBO(num_particles=K): K guide draws in sequence (pyro draws the particles one after
        the other), each through the whole ELBO + gradient kernel sequence; the K losses and gradients are averaged, then one
        all-reduce (cells sharded) and one optimiser step."""
        e, K = self.e, self.K
        acc = torch.zeros_like(e.grad)
        lacc = torch.zeros((), dtype=torch.float64, device=e.grad.device)
        for k in range(K):
            if eps_list is not None:
                eps = eps_list[k]
            elif getattr(self, "_pending", None) is not None:
                eps, self._pending = self._pending, None
            else:
                eps = draw_eps(e.spec, self.gen)
            e.elbo_grad(eps=e.pack_eps(eps), step=self.step_idx * K + k)
            acc += e.grad
            lacc += e.grad[:2].double().sum()          # the loss header is a float hi / lo pair: averaged in float64
        e.grad.copy_(acc / K)
        self._put_loss_header(lacc / K)
        self._reduce()
        loss = float(e.grad[:2].double().sum().item())
        self._update()
        self.step_idx += 1
        self.losses.append(loss)
        return loss

    def _put_loss_header(self, loss64: torch.Tensor):
        """grad[0:2] = the float hi / lo split of a float64 loss (what K_fin writes): hi / K rounded on its own would lose
        ~6e-8 |hi| of the double-float pair."""
        hi = loss64.float()
        self.e.grad[0] = hi
        self.e.grad[1] = (loss64 - hi.double()).float()

    def _perf_particles(self, n_steps: int):
        """perf mode with K particles: per step the unfused kernel sequence on the Philox streams (seed, step * K + k),
        averaged on the device; losses stay in the device ring."""
        e, K = self.e, self.K
        if not self.do_reduce and self.opt.impl == "hip" and not self.tuning.particles_host_loop:
            # single rank: every launch of the run from one C call (vc_svi_run_particles) -- the same kernels on the same
            # streams, the gradients averaged by a kernel instead of PyTorch ops: the same numbers
            if getattr(self, "_gacc", None) is None:
                self._gacc = torch.zeros_like(e.grad)
            o = self.opt
            e.svi_run_particles(self._gacc, o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, seed=self.seed,
                                step_dev=self.step_dev, step0=self.step_idx, num_particles=K, n_steps=n_steps,
                                loss_buf=self.loss_hist)
            o.t += n_steps
            self.step_idx += n_steps
            return
        for _ in range(n_steps):
            acc = torch.zeros_like(e.grad)
            lacc = torch.zeros((), dtype=torch.float64, device=e.grad.device)
            for k in range(K):
                e.elbo_grad(eps=None, seed=self.seed, step=self.step_idx * K + k)
                acc += e.grad
                lacc += e.grad[:2].double().sum()
            e.grad.copy_(acc / K)
            self._put_loss_header(lacc / K)
            self._reduce()
            self.loss_hist[self.step_idx] = e.grad[:2].double().sum()
            self._update()
            self.step_idx += 1
            self.step_dev += 1                 # the device counter mirrors the steps done (checkpoints read it)

    def invalidate(self):
        """Tell the runner that params / step counter / seed were changed from outside (fused3 keeps the NEXT step's
        sample in the engine's tables; it is re-drawn from the current parameters before the next step)."""
        self._primed = False

    def _perf_body(self, prime: bool = False, n_steps: int = 1):
        e = self.e
        if self.adam_impl == "sharded":        # cells sharded: K_main -> phase A -> sum over ranks -> phase B
            from . import _lib
            o = self.opt
            kw = dict(seed=self.seed, step_dev=self.step_dev, loss_buf=self.loss_hist)
            if (self.exchange == "engine" and (self.world > 1 or (self.do_reduce and self.tuning.exchange_check))
                    and not getattr(self, "_exchange_checked", False) and n_steps > 0):
                # The first step of the engine-owned exchange is cut open once: phase A -> the buffer summed by the engine's
                # communicator AND, on a copy, by torch.distributed -> compared -> phase B.  If any
                # rank sees a difference every rank falls back to the torch exchange (MIN all-reduce of the verdict).
                import torch.distributed as dist
                self._exchange_checked = True
                e.svi_run_sharded(self.xbuf, o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, prime=prime,
                                  phase=_lib.VC_PHASE_A, **kw)
                ref = self.xbuf.clone()
                mag = self.xbuf.abs()
                dist.all_reduce(ref, group=self.pg)
                dist.all_reduce(mag, group=self.pg)
                # two communicators never have a collective in flight at the same time (their kernels could be scheduled in
                # different orders on different ranks): torch's are complete on every rank before the engine's is enqueued
                torch.cuda.synchronize(e.device)
                e.comm_allreduce(self.xbuf)
                torch.cuda.synchronize(e.device)
                # tolerance relative to the sum of the ranks' MAGNITUDES: two correct all-reduces may add the ranks in different
                # orders, and a gradient whose terms cancel differs by ~1e-7 of those terms, not of the result
                diff = (self.xbuf - ref).abs()
                same = bool(((diff <= 1e-5 * mag + 1e-30) | (torch.isnan(self.xbuf) & torch.isnan(ref))).all().item())
                flag = torch.tensor([1 if same else 0], dtype=torch.int32, device=e.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
                self.exchange_check = "ok" if int(flag.item()) else "mismatch"
                if not int(flag.item()):
                    import warnings
                    warnings.warn("the engine's RCCL communicator did not reproduce torch.distributed's sum of the exchange "
                                  "buffer; every rank continues with the torch exchange")
                    self.exchange = "torch"
                    self.xbuf.copy_(ref)
                e.svi_run_sharded(self.xbuf, o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, phase=_lib.VC_PHASE_B, **kw)
                prime, n_steps = False, n_steps - 1
                if n_steps == 0:
                    return
            if self.exchange in ("engine", "p2p", "none"):      # every launch and every all-reduce of the run from one C call
                e.svi_run_sharded(self.xbuf, o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, prime=prime,
                                  phase=_lib.VC_PHASE_AB, n_steps=n_steps, **kw)
                return
            import torch.distributed as dist
            for i in range(n_steps):
                e.svi_run_sharded(self.xbuf, o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, prime=prime and i == 0,
                                  phase=_lib.VC_PHASE_A, **kw)
                dist.all_reduce(self.xbuf, group=self.pg)
                e.svi_run_sharded(self.xbuf, o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, phase=_lib.VC_PHASE_B, **kw)
            return
        if self.adam_impl == "fused3":         # single rank: K_main -> K_tail -> K_omega
            o = self.opt
            e.svi_step_fused(o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, seed=self.seed,
                             step_dev=self.step_dev, loss_buf=self.loss_hist, prime=prime, n_steps=n_steps)
            return
        if self.adam_impl == "fused":          # single rank: optimiser merged into the last gradient kernel
            o = self.opt
            e.svi_step(o.m, o.v, o.lr0, o.lrd, o.b1, o.b2, o.eps, o.clip, eps=None, seed=self.seed, step=0,
                       step_dev=self.step_dev, loss_buf=self.loss_hist)
            return
        # K_fin writes the loss into slot step % len(loss_hist); K_post has advanced step_dev
        e.elbo_grad(eps=None, seed=self.seed, step=0, step_dev=self.step_dev, loss_buf=self.loss_hist)
        if self.do_reduce:
            self._reduce()
            if self.opt.impl == "hip":     # the optimiser kernel files the reduced loss (no extra launches)
                self._update(t_dev=self.step_dev, loss_hdr=e.grad, loss_ring=self.loss_hist)
                return
            idx = (self.step_dev - 1) % self.loss_hist.shape[0]
            self.loss_hist.index_copy_(0, idx, e.grad[:2].double().sum().reshape(1))
        self._update(t_dev=self.step_dev)          # step_dev now holds the 1-based Adam step

    def run_perf(self, n_steps: int, sync: bool = True) -> None:
        """n_steps back-to-back SVI steps with no host round trip (losses stay on the device)."""
        e = self.e
        if n_steps <= 0:          # nothing to do: in particular no graph warm-up pass, which is one real step
            return
        if self.loss_hist is None or self.loss_hist.shape[0] < self.step_idx + n_steps:
            # (a captured graph holds the ring's address: start large enough that a long run does not have to re-capture)
            new = torch.zeros(max(2 * (self.step_idx + n_steps), 16384), dtype=torch.float64, device=e.device)
            if self.loss_hist is not None:
                new[: self.loss_hist.shape[0]] = self.loss_hist
            self.loss_hist = new
            self._graph = None
        if self.K > 1:
            self._perf_particles(n_steps)
            if sync:
                torch.cuda.synchronize(e.device)
            return
        if self.use_graph and self._graph is None:
            s = torch.cuda.Stream(device=e.device)
            s.wait_stream(torch.cuda.current_stream(e.device))
            g = None
            with torch.cuda.stream(s):
                self._perf_body(prime=not self._primed)   # warm-up (allocator, lazy init, communicator) outside capture
                self._primed = True
                torch.cuda.synchronize()
                # ProcessGroupNCCL's watchdog thread polls the events of the collectives issued before the capture;
                # under the default "global" capture mode such a query from ANOTHER thread invalidates the capture on
                # ROCm (hipErrorStreamCaptureUnsafe).  "thread_local" restricts the check to the capturing thread, which
                # issues only capturable work here -- deterministic, no waiting for the watchdog to go idle.
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local" if self.do_reduce else "global"):
                        self._perf_body()
                except Exception as ex:                  # capture refused (driver / RCCL build): plain launches, same results
                    import warnings
                    warnings.warn(f"hipGraph capture of the SVI step failed ({type(ex).__name__}: {ex}); using eager launches")
                    g = None
                    torch.cuda.synchronize()
                if self.do_reduce and self.world > 1:
                    # every rank replays or none does: a rank that fell back to eager launches beside ranks that replay a
                    # captured collective would issue its all-reduces in a different order
                    import torch.distributed as dist
                    flag = torch.tensor([0 if g is None else 1], dtype=torch.int32, device=e.device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
                    if int(flag.item()) == 0:
                        g = None
                if g is None:
                    self.use_graph = False
            torch.cuda.current_stream(e.device).wait_stream(s)
            # the warm-up pass was one real step (capture only records)
            self._graph = g
            self.step_idx += 1
            n_steps -= 1
        if self._graph is None and self.adam_impl in ("fused3", "sharded"):
            self._perf_body(prime=not self._primed, n_steps=n_steps)      # every launch of the run from one C call
            self._primed = True
        else:
            for _ in range(n_steps):
                if self._graph is not None and self._primed:
                    self._graph.replay()
                else:
                    self._perf_body(prime=not self._primed)
                    self._primed = True
        self.step_idx += n_steps
        if sync and torch.device(e.device).type == "cuda":
            self._bounded_sync()

    def _bounded_sync(self):
        """torch.cuda.synchronize with a deadline when cells are sharded: a stuck collective or a dead peer otherwise hangs every
        rank for as long as its caller is willing to wait (bench.py has its own watchdog; fit() / run_svi had none).  The
        stream is polled through an event; past VC_RUN_DEADLINE_S (default 900 s) the run is given up with an exception --
        the process is never re-executed, the device work is not cancelled (exit the process to release it)."""
        e = self.e
        if self.world <= 1 and not self.do_reduce:
            torch.cuda.synchronize(e.device)
            return
        import time
        limit = float(self.tuning.run_deadline_s)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(e.device))
        t0 = time.monotonic()
        while not ev.query():
            waited = time.monotonic() - t0
            if waited > limit:
                from .engine import HipEngineError
                raise HipEngineError(f"rank {e.rank} of {self.world}: the sharded run did not finish within {limit:.0f} s "
                                     f"(Tuning.run_deadline_s) -- a collective or a peer is stuck (exchange: {self.exchange})")
            # the return must not lag the device by more than ~0.1 % of the run (callers time run_perf(sync=True)): spin for the
            # first quarter second, then naps of a thousandth of what has been waited so far (at most 2 ms)
            if waited > 0.25:
                time.sleep(min(waited * 1e-3, 2e-3))

    _SENTINEL = -0x0007_2174_5EED_0001          # an int64 bit pattern no loss takes (a NaN with this payload)

    def step_with_loss(self) -> float:
        """One perf-mode step that hands its loss to the host, as `svi.step()` does (velocity_inference_model.py:118-121),
        without a stream synchronise or a device-to-host copy: the loss ring is moved to pinned host memory, which the
        device writes directly (K_omega's loss block), and the host spins on the slot of this step.  Single-rank fused
        path only; every other configuration runs the step and copies the loss back."""
        e = self.e
        if self.loss_every > 1:
            raise ValueError("step_with_loss hands over the loss of every step: not with loss_every > 1")
        if self.mode != "perf" or self.adam_impl != "fused3" or self._graph is not None or self.use_graph:
            self.run_perf(1, sync=False)
            return float(self.loss_hist[self.step_idx - 1].item())
        import numpy as np
        ring = self.loss_hist
        if ring is None or ring.is_cuda or ring.shape[0] < self.step_idx + 1:
            n = max(2 * (self.step_idx + 1), 16384)
            new = torch.zeros(n, dtype=torch.float64).pin_memory()
            if ring is not None:
                torch.cuda.synchronize(e.device)
                new[: min(ring.shape[0], n)] = ring[: min(ring.shape[0], n)].cpu()
            self.loss_hist = ring = new
            self._ring_i64 = ring.numpy().view(np.int64)
        i64 = self._ring_i64
        k = self.step_idx % ring.shape[0]
        i64[k] = self._SENTINEL
        self._perf_body(prime=not self._primed, n_steps=1)
        self._primed = True
        self.step_idx += 1
        spins = 0
        while i64[k] == self._SENTINEL:
            spins += 1
            if spins > 2_000_000:                       # a step takes ~0.1 ms: something failed -- surface the error
                torch.cuda.synchronize(e.device)
                if i64[k] == self._SENTINEL:
                    raise RuntimeError("the device never wrote the loss of this step")
        return float(ring[k])

    def perf_losses(self) -> List[float]:
        if torch.device(self.e.device).type == "cuda":
            torch.cuda.synchronize(self.e.device)
        if self.loss_hist is None:
            return []
        out = self.loss_hist[: self.step_idx].cpu().tolist()
        if self.loss_every > 1:          # steps that ran the gradient-only kernel have no loss
            base = getattr(self, "_loss_base", 0)
            out = [x if i < base or (i - base) % self.loss_every == 0 else float("nan") for i, x in enumerate(out)]
        return out

    def _arm_loss_every(self):
        """(Re)starts the engine's count of likelihood launches at the runner's current step: step base + j k forms the loss."""
        if self.loss_every > 1:
            if self.mode != "perf" or self.adam_impl != "fused3" or self.K != 1 or self.use_graph:
                # (a captured graph would replay whichever kernel its one captured step chose)
                raise ValueError("loss_every > 1 needs the fused single-rank perf-mode step (one particle, no hipGraph replay)")
            self.e.set_loss_every(self.loss_every)
            self._loss_base = self.step_idx
        elif hasattr(self.e, "set_loss_every") and self.mode == "perf" and self.adam_impl == "fused3":
            self.e.set_loss_every(1)        # the engine outlives its runners: an earlier runner's period must not linger

    # ------------------------------------------------------------------------------------------
    # checkpoint / resume of a fit (the reference's analogue: pyro.get_param_store().get_state()/set_state(),
    # tutorials/1D_Pancreas_Analysis.ipynb cell 26; here the optimiser moments, the step counter and the RNG position
    # are part of the snapshot, so a resumed run continues the SAME trajectory bit for bit)
    def _layout_key(self) -> str:
        e = self.e
        return repr((e.spec.kind, e.spec.guide, e.spec.noisemodel, e.header, e.n_global, e.n_local, e.rank,
                     e.world_size, sorted(e.param_slices.items())))

    def state_dict(self) -> dict:
        """Everything needed to continue this run: flat unconstrained parameters (this rank's block of ϕxy_locs
        included), ClippedAdam moments, steps done, seed / host generator state, the losses so far."""
        e = self.e
        torch.cuda.synchronize(e.device)
        opt = self.opt.state_dict(self.step_dev if self.mode == "perf" else None)
        sd = dict(layout=self._layout_key(), mode=self.mode, seed=self.seed, step_idx=self.step_idx,
                  params=e.params.detach().cpu().clone(), m=opt["m"], v=opt["v"], t=opt["t"], **({"t_vec": opt["t_vec"]} if "t_vec" in opt else {}),
                  losses=torch.tensor(self.perf_losses() if self.mode == "perf" else self.losses, dtype=torch.float64))
        if self.gen is not None:
            sd["gen_state"] = self.gen.get_state()
        return sd

    def load_state_dict(self, sd: dict) -> None:
        e = self.e
        if sd["layout"] != self._layout_key():
            raise ValueError("checkpoint does not match this engine's parameter layout / rank")
        if sd["mode"] != self.mode:
            raise ValueError(f"checkpoint was written in mode={sd['mode']!r}, this runner is mode={self.mode!r}")
        e.params.copy_(torch.as_tensor(sd["params"]).to(e.device))
        self.opt.load_state_dict(dict(m=sd["m"], v=sd["v"], t=sd["t"], t_vec=sd.get("t_vec")))
        self.step_idx, self.seed = int(sd["step_idx"]), int(sd["seed"])
        losses = torch.as_tensor(sd["losses"], dtype=torch.float64)
        self._primed = False
        self._arm_loss_every()
        if self.mode == "perf":
            self.step_dev.fill_(self.step_idx)
            n = max(2 * self.step_idx, 16384)
            if self.loss_hist is None or self.loss_hist.shape[0] < n:
                self.loss_hist = torch.zeros(n, dtype=torch.float64, device=e.device)
                self._graph = None          # the captured graph holds the old ring's address
            self.loss_hist[: losses.numel()] = losses.to(e.device)
        else:
            self.losses = losses.tolist()
            self._first, self._pending = None, None
            if "gen_state" in sd and self.gen is not None:
                self.gen.set_state(sd["gen_state"])

    def save(self, path: str) -> None:
        import numpy as np
        sd = self.state_dict()
        np.savez(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()})

    def load(self, path: str) -> None:
        import numpy as np
        if not path.endswith(".npz"):
            path += ".npz"
        z = np.load(path, allow_pickle=False)
        sd = {k: z[k] for k in z.files}
        for k in ("layout", "mode"):
            sd[k] = str(sd[k])
        for k in ("seed", "step_idx", "t"):
            sd[k] = int(sd[k])
        for k in ("params", "m", "v", "losses", "gen_state", "t_vec"):
            if k in sd:
                sd[k] = torch.from_numpy(np.array(sd[k]))
        self.load_state_dict(sd)
