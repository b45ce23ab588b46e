"""`PhaseFitModel` / `VelocityFitModel`: the reference's fit drivers (phase_inference_model.py:81-341,
velocity_inference_model.py:32-302) on top of the HIP engine.  Same constructor and `fit()` signature,
same attributes after the fit (`losses`, `phis_pyro`, `fourier_coef`, `fourier_coef_sd`, `disp_pyro`,
`delta_nus`, `cycle_pyro`, `phase_pyro`, `log_gammas`, `log_betas`, `velocity_coef[_sd]`, `speed_pyro`,
`posterior`, `metaparams_avg`).  Extra keyword arguments of `fit()` (not in the reference):
  mode   "perf" (default): eps from the engine's Philox stream (seeded from torch.initial_seed()), whole
         step replayed from a hipGraph, losses read back at the end;
         "parity": eps drawn on the host from torch's default generator in Pyro's order, so that
         `torch.manual_seed(s); fit(...)` reproduces the reference run with the same seed.
  process_group   cells sharded over the ranks of a `torch.distributed` group (default: the world group when
         torch.distributed is initialised): every rank passes the SAME full-size metaparams, keeps
         its contiguous block of cells on its GPU, the step all-reduces the gene-level gradients once, and at the
         end the per-cell results are gathered so that every rank holds the same full-`Nc` attributes as a
         single-process fit (shard-count invariant: same Philox / host eps streams, sliced by cell offset).
"""
from __future__ import annotations

import copy
import logging
import math
from typing import Dict, Optional

import numpy as np
import torch

from . import pyro_compat
from .containers import AngularSpeed, Cycle, Phases
from .distributed import broadcast_int, dist_context, gather_cells
from .engine import HipEngine, shard_bounds
from .spec import ModelSpec, spec_from_metaparams
from .svi import SVIRunner, optim_args_of
from .utils import torch_basis, torch_fourier_basis


def _perf_seed(seed: int, drawn: int, counter_continues: bool, continues_store: bool) -> int:
    """Philox key of a perf-mode fit.  The step counter of the runner is Philox step, loss slot and ClippedAdam step in one: a fit
    that continues from the param store with the SAME optimizer object carries that counter on (a fresh part of the stream
    (seed, t0 ...): fit(n) + fit(n) == fit(2n)); a continued fit whose counter restarts at 0 -- a new optimizer object, a dict,
    mixed per-parameter step counts -- would draw (seed, 0 ...) again, the very noise of the fit it continues, whereas Pyro's
    global RNG keeps advancing.  Such a fit takes a key derived from (seed, perf steps drawn since the store was cleared)."""
    if counter_continues or not continues_store or drawn <= 0:
        return int(seed)
    x = (int(seed) ^ ((int(drawn) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF            # splitmix64 finaliser
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return int((x ^ (x >> 31)) % (2 ** 63))


class _FitBase:
    _kind = None
    # The reference's default `loss=Trace_ELBO(...)` object is created once per class and shared by later
    # fit() calls, so only the FIRST fit of a process makes Trace_ELBO's extra guide pass (one eps set drawn
    # and discarded before step 0).  Mirrored per class for mode="parity".
    _default_elbo_fresh = True

    def __init__(self, metaparams, condition_on={}, early_exit=False, get_posterior=True, num_samples=500,
                 n_per_bin=50):
        self.model = metaparams.model_fn
        self.guide = metaparams.guide_fn
        self.posterior = None
        self.condition = condition_on
        self.condition_on = list(condition_on.keys())
        self.metaparams = metaparams
        self.early_exit = early_exit
        self.get_posterior = get_posterior
        self.num_samples = num_samples
        self.n_per_bin = n_per_bin
        self.engine: Optional[HipEngine] = None

    # ------------------------------------------------------------------------------------------
    def _make_engine(self, device=None, process_group=None, tuning=None):
        spec = spec_from_metaparams(self.metaparams, self._kind, self.condition)
        dev = device if device is not None else getattr(self.metaparams, "device", None)
        if dev is None or torch.device(dev).type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.spec = spec
        self._rank, self._world, self._pg = dist_context(process_group)
        self._shard_sizes = [b - a for a, b in (shard_bounds(spec.Nc, r, self._world) for r in range(self._world))]
        # raises without a GPU: no CPU fallback
        self.engine = HipEngine(spec, device=dev, rank=self._rank, world_size=self._world, tuning=tuning)
        return self.engine

    def _gather(self, local: torch.Tensor, dim: int) -> torch.Tensor:
        """This rank's block of cells along `dim` -> the full-Nc CPU tensor, on every rank."""
        return gather_cells(local, dim, self._shard_sizes, self._pg)

    def _warn_sizes(self):
        pass

    def fit(self, optimizer, loss=None, num_steps=1000, intermediate_output_step_size=100, store_output=False,
            verbose=True, mode: str = "perf", seed: Optional[int] = None, device=None, process_group=None, loss_every: int = 1,
            tuning=None):
        """The reference's fit() (velocity_inference_model.py:77-160, phase_inference_model.py:126-214) + mode / seed / device /
        process_group of this engine.  loss_every = k > 1 is an opt-in that is NOT in the reference: only every k-th step forms
        the loss (`losses` holds NaN in between), the others run a gradient-only likelihood kernel -- NB noise, perf mode, one
        rank; + 23 % steps/s for the velocity stage of the tutorial flow (phases, nu, delta nu, shape_inv conditioned), + 4-7 % for
        the models that learn shape_inv.  tuning: a `velocycle_amd.tuning.Tuning` (engine knobs as DATA: the library reads no
        environment variable); None = the measured defaults.  The engine is built by the first fit() and kept: a later call with
        another Tuning raises."""
        self._warn_sizes()
        import time
        t_start = time.perf_counter()
        if self.engine is None:
            self._make_engine(device, process_group, tuning)
        elif tuning is not None and tuning != self.engine.tuning:
            raise ValueError("this model's engine was built with another Tuning (the engine is kept across fit() calls); "
                             "make a new model object for a new tuning")
        eng = self.engine
        eng.clear_status()           # the engine is kept across fit() calls: the NaN / Inf latch must describe THIS fit only
        if eng.stats["count_storage"] != "u16" and self.spec.noisemodel != "Lognormal":
            import warnings
            warnings.warn("velocycle_amd: this rank's count matrices hold a non-integer value or a count > 65535, so they are "
                          "kept as float32 in HBM (twice the bytes per step of the uint16 layout; results are unaffected)",
                          RuntimeWarning)
        torch.cuda.synchronize(eng.device)
        t_engine = time.perf_counter()
        args = optim_args_of(optimizer)
        # the user's ELBO object (velocity_inference_model.py:79,111): num_particles = K guide draws per step, averaged
        particles = int(getattr(loss, "num_particles", 1)) if loss is not None else 1
        if particles < 1:
            raise ValueError("loss.num_particles must be >= 1")
        if loss is not None and getattr(loss, "vectorize_particles", False) and mode == "parity" and particles > 1:
            # the same estimator (K draws averaged): perf mode runs it as the batched particle step.  Parity mode promises the
            # reference's host RNG order, and a vectorised trace consumes the generator site-major -- an order no fixture pins
            # (pyro_compat.Trace_ELBO's docstring)
            raise NotImplementedError("Trace_ELBO(vectorize_particles=True) with mode='parity': the host RNG order of a vectorised "
                                      "trace is not pinned; use mode='perf' (same estimator: K draws per step, averaged)")
        exact = mode == "parity" or self.early_exit or store_output
        if int(loss_every) > 1 and (exact or particles > 1 or self._world > 1):
            raise ValueError("loss_every > 1 needs mode='perf' on one rank without early_exit / store_output / particles "
                             "(they read the loss of every step)")
        # continue-from-store (pyro_compat): parameters the store still holds are the starting values, and an optimizer OBJECT
        # that stepped on them before carries its moments / step counts on
        plan = self._continuation_plan(optimizer)
        if mode == "parity":
            gen_seed = seed
            if gen_seed is None and self._world > 1:
                # sharded: every rank must draw the same host eps stream -> rank 0's global seed, as an explicit
                # generator (equal to the global RNG right after torch.manual_seed(s))
                gen_seed = broadcast_int(torch.initial_seed() % (2 ** 63), self._pg, eng.device)
            warm = type(self)._default_elbo_fresh if loss is None else bool(getattr(loss, "fresh", True))
            run = SVIRunner(eng, args, mode="parity", seed=gen_seed, warmup_draw=warm,   # seed None -> torch's global RNG
                            process_group=self._pg, num_particles=particles)
            if loss is None:
                type(self)._default_elbo_fresh = False
            elif hasattr(loss, "fresh"):
                try:
                    loss.fresh = False       # the object has made its extra guide pass: a later fit with it will not
                except Exception:
                    pass
        else:
            s = int(torch.initial_seed() % (2 ** 63)) if seed is None else int(seed)
            # (a continued fit whose step counter restarts must not replay the first fit's eps sequence)
            s = _perf_seed(s, pyro_compat.perf_steps_drawn(), (not plan["mixed"]) and int(plan["t0"]) > 0, bool(plan["values"]))
            s = broadcast_int(s, self._pg, eng.device)
            if int(loss_every) > 1 and plan["mixed"]:
                raise ValueError("loss_every > 1 cannot be combined with a fit that continues parameters of DIFFERENT step counts "
                                 "(the same optimizer object after a phase fit and a velocity fit): that continuation runs the "
                                 "per-element-step ClippedAdam, not the fused step that has the gradient-only kernel")
            run = SVIRunner(eng, args, mode="perf", seed=s, process_group=self._pg, num_particles=particles,
                            adam_impl="torch" if plan["mixed"] else None, loss_every=int(loss_every))
        self._runner = run
        t_first = self._apply_continuation(run, plan)
        if int(loss_every) > 1:
            run._arm_loss_every()            # (the continuation moved the step counter: the period starts at THIS fit's first step)
        losses, intermediate_output = [], []
        if mode == "perf" and not exact:
            run.run_perf(num_steps)
            losses = run.perf_losses()[t_first:]
        else:
            early = False
            for step in range(num_steps):
                if mode == "parity":
                    l = run.step()
                else:
                    l = run.step_with_loss()
                losses.append(l)
                if store_output and step % intermediate_output_step_size == 0:
                    logging.info("Elbo loss: {}".format(l))
                    intermediate_output.append(self.sample_posterior(num_samples=self.n_per_bin))
                if early:                                    # velocity_inference_model.py:147-151
                    if np.abs(np.mean(losses[-100:]) - np.mean(losses[-10:])) < 5:
                        break
                elif step > 200 and self.early_exit:
                    early = True
        self.losses = losses
        if mode == "perf":
            pyro_compat.add_perf_steps(len(losses))
        torch.cuda.synchronize(eng.device)
        t_svi = time.perf_counter()
        ok, first_bad, n_bad = eng.status()                                     # device-side latch of the C ABI
        formed = np.asarray(losses, dtype=np.float64)[:: max(1, int(loss_every))]        # (loss_every > 1: NaN in between by design)
        if not ok or not np.all(np.isfinite(formed)):   # pyro.util.warn_if_nan(loss, "loss")
            import warnings
            warnings.warn("Encountered NaN/Inf: loss" + ("" if ok else f" (first at step {first_bad}, {n_bad} steps)"),
                          UserWarning)
        self._extract()
        self._remember_optimizer_state(optimizer, run, plan, len(losses))
        t_extract = time.perf_counter()
        if self.get_posterior:
            self._posterior()
        # wall seconds of the stages of this call (not in the reference; profiles/tools/fit_wall_time.py prints them)
        self.timings = {"engine_setup": t_engine - t_start, "svi_steps": t_svi - t_engine, "extract": t_extract - t_svi,
                        "posterior": time.perf_counter() - t_extract, "count_storage": eng.stats["count_storage"]}
        if store_output:
            return intermediate_output

    # ------------------------------------------------------------------------------------------
    _PYRO_SHAPES = staticmethod(lambda sp: {
        "ν_locs": (sp.Ng, 1, sp.Nh), "ν_scales": (sp.Ng, 1, sp.Nh), "ϕxy_locs": (sp.Nc, 2),
        "shape_inv_locs": (sp.Ng, 1), "logγg_locs": (sp.Ng, 1), "logγg_scales": (sp.Ng, 1),
        "logβg_locs": (sp.Ng, 1), "logβg_scales": (sp.Ng, 1), "νω_locs": (sp.Nx, sp.Nhw, 1, 1),
        "νω_scales": (sp.Nx, sp.Nhw, 1, 1),
        "Δν_locs": (sp.Nb, sp.Ng, 1) if sp.kind == "phase" else (sp.Nb, 1, 1, sp.Ng, 1)})

    def _unconstrained_host(self) -> Dict[str, torch.Tensor]:
        """The engine's parameters as CPU tensors of the full problem (ϕxy_locs, the only rank-local block, gathered
        once: SURVEY.md §8e), unconstrained as Pyro stores them."""
        return {k: (self._gather(v, 0) if k == "ϕxy_locs" else v.detach().cpu().clone()) for k, v in self.engine.named().items()}

    def _constrained(self) -> Dict[str, torch.Tensor]:
        from ._lib import POSITIVE_PARAMS
        self._raw_params = self._unconstrained_host()
        return {k: (v.exp() if k in POSITIVE_PARAMS else v) for k, v in self._raw_params.items()}

    def _publish(self, par):
        """Fill the param store with Pyro's names and shapes (and, beside it, the raw values a later fit() continues from)."""
        shp = self._PYRO_SHAPES(self.spec)
        for k, v in par.items():
            pyro_compat.publish(k, v.reshape(shp.get(k, v.shape)), self._raw_params[k])

    # ---- continue-from-store (reference: velocity_inference_guide.py:25-43, phase_inference_guide.py:36-45) -------------
    def _continuation_plan(self, optimizer) -> dict:
        """Which of this model's parameters the param store still holds (they are the starting values of this fit, as
        `pyro.param(name, init)` returns the stored value), and what the optimizer object remembers about them: PyroOptim
        keeps one torch optimiser per parameter TENSOR, so the same optimizer object passed again carries moments, step
        count and decayed learning rate on, a new object (or a cleared store: new tensors) starts afresh."""
        from ._lib import POSITIVE_PARAMS
        eng, sp = self.engine, self.spec
        values, steps = {}, {}
        state = getattr(optimizer, "_vc_state", None) if not isinstance(optimizer, dict) else None
        if state is not None and state.get("generation") != pyro_compat.generation():
            state = None
        for name in eng.param_slices:
            u = pyro_compat.stored_unconstrained(name, name in POSITIVE_PARAMS)
            steps[name] = 0
            if u is None:
                continue
            full = (sp.Nc, 2) if name == "ϕxy_locs" else eng.param_shape(name)
            want = 1
            for n in full:
                want *= int(n)
            if u.numel() != want:
                raise RuntimeError(
                    f"pyro.param({name!r}) in the param store holds {u.numel()} values, this model's guide needs {want} "
                    f"{tuple(full)}: the store belongs to another model / data set -- call pyro.clear_param_store() before this fit()")
            values[name] = u.reshape(full)
            if state is not None and name in state["names"]:
                steps[name] = int(state["names"][name]["t"])
        ts = set(steps.values())
        return {"values": values, "steps": steps, "state": state, "mixed": len(ts) > 1, "t0": (ts.pop() if len(ts) == 1 else 0)}

    def _apply_continuation(self, run, plan) -> int:
        """Puts the stored parameters / optimiser state into the freshly initialised runner; returns the index of this fit's
        first step in the runner's counters (perf mode: Philox step, loss ring slot and ClippedAdam step count are one
        device counter)."""
        eng = self.engine
        if not plan["values"]:
            return 0
        eng.set_params(plan["values"])
        run.invalidate()
        st, h = plan["state"], eng.header
        if st is not None:
            for name, (off, n) in eng.param_slices.items():
                ent = st["names"].get(name)
                if ent is None or name not in plan["values"]:
                    continue
                m, v = ent["m"], ent["v"]
                if name == "ϕxy_locs" and m.shape[0] != eng.Nc_local:
                    m, v = m[eng.c0:eng.c1], v[eng.c0:eng.c1]
                run.opt.m[off - h:off - h + n].copy_(m.reshape(-1).to(eng.device))
                run.opt.v[off - h:off - h + n].copy_(v.reshape(-1).to(eng.device))
        if plan["mixed"]:
            tv = torch.zeros(eng.total - h, dtype=torch.float64, device=eng.device)
            for name, (off, n) in eng.param_slices.items():
                tv[off - h:off - h + n] = float(plan["steps"][name])
            run.opt.set_step_counts(tv)
            return 0
        t0 = int(plan["t0"])
        run.opt.t = t0
        if run.mode == "perf":
            run.step_idx = t0
            run.step_dev.fill_(t0)
        return t0 if run.mode == "perf" else 0

    def _remember_optimizer_state(self, optimizer, run, plan, steps_done: int):
        """After the fit: per parameter name the moments and the step count, kept ON the optimizer object (dict arguments
        have no identity to carry state)."""
        if isinstance(optimizer, dict):
            return
        eng, h = self.engine, self.engine.header
        st = plan["state"] or {"generation": pyro_compat.generation(), "names": {}}
        for name, (off, n) in eng.param_slices.items():
            m = run.opt.m[off - h:off - h + n].reshape(eng.param_shape(name))
            v = run.opt.v[off - h:off - h + n].reshape(eng.param_shape(name))
            if name == "ϕxy_locs":
                m, v = self._gather(m, 0), self._gather(v, 0)
            else:
                m, v = m.detach().cpu().clone(), v.detach().cpu().clone()
            st["names"][name] = {"m": m, "v": v, "t": plan["steps"][name] + int(steps_done)}
        try:
            optimizer._vc_state = st
        except Exception:
            pass

    def _extract(self):
        sp = self.spec
        par = self._constrained()
        self._publish(par)
        P = pyro_compat._STORE
        self.phis_pyro = P["ϕxy_locs"].squeeze().numpy().T
        self.fourier_coef = P["ν_locs"].squeeze().numpy().T
        self.fourier_coef_sd = P["ν_scales"].squeeze().numpy().T
        new_cycle = Cycle.from_array(self.fourier_coef, self.fourier_coef_sd, self.metaparams.cycle_prior.genes)
        new_phase = Phases.from_array(self.phis_pyro, cell_names=self.metaparams.phase_prior.phi_xy.columns)
        if "shape_inv_locs" in P:
            self.disp_pyro = P["shape_inv_locs"].squeeze().numpy().T
        if sp.with_delta_nu:
            d = P["Δν_locs"]
            self.delta_nus = (d.unsqueeze(-3).unsqueeze(-4) if sp.kind == "phase" else
                              d.unsqueeze(-3).unsqueeze(-4)).float().numpy()
        self.cycle_pyro, self.phase_pyro = new_cycle, new_phase
        return par

    def _draw_sites(self, n, rs=None) -> Dict[str, torch.Tensor]:
        """n guide draws pushed through the deterministic part of the model, as DEVICE tensors with Pyro's site
        shapes (leading dimension n).  One library call (vc_sample_posterior) makes all n draws."""
        eng, sp = self.engine, self.spec
        base = broadcast_int(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()), self._pg, eng.device)
        names = [k for k in ("ν", "Δν", "ϕxy", "shape_inv", "logγg", "logβg", "νω", "rho_real")
                 if self._site_exists(k)] + ["ϕ"] + (["ω"] if sp.kind == "velocity" else [])
        dev = eng.sample_posterior(names, n, seed=base, step0=0)
        if self._world > 1:
            # gene-level / global sites are identical on every rank (same Philox key); per-cell sites are this
            # rank's block of the global stream: gather them (cells are dimension 1 of the (n, Nc_local, ...) draws)
            for k in ("ϕxy", "ϕ", "ω"):
                if k in dev:
                    dev[k] = self._gather(dev[k], 1)
            dev = {k: v.cpu() for k, v in dev.items()}
        want = (lambda k: True) if rs is None else (lambda k: k in rs)

        def basis(num_harmonics, der):          # torch_fourier_basis over every draw at once, (n, Nc, Nh)
            return torch_fourier_basis(dev["ϕ"].reshape(-1), num_harmonics, der=der,
                                       device=dev["ϕ"].device).reshape(n, sp.Nc, -1)

        res = {}
        if want("ν"): res["ν"] = dev["ν"].reshape(n, sp.Ng, 1, sp.Nh)
        if want("ϕxy"): res["ϕxy"] = dev["ϕxy"].reshape(n, sp.Nc, 2)
        if want("ϕ"): res["ϕ"] = dev["ϕ"]
        if want("ζ"): res["ζ"] = basis(sp.H, 0)
        if "shape_inv" in dev and want("shape_inv"):
            res["shape_inv"] = dev["shape_inv"].reshape(n, sp.Ng, 1)
        if "Δν" in dev and want("Δν"):
            res["Δν"] = dev["Δν"].reshape((n, sp.Nb, sp.Ng, 1) if sp.kind == "phase" else (n, sp.Nb, 1, 1, sp.Ng, 1))
        if sp.kind == "velocity":
            if want("logγg"): res["logγg"] = dev["logγg"].reshape(n, sp.Ng, 1)
            if want("γg"): res["γg"] = dev["logγg"].reshape(n, sp.Ng, 1).exp()
            if want("logβg"): res["logβg"] = dev["logβg"].reshape(n, sp.Ng, 1)
            if want("νω"): res["νω"] = dev["νω"].reshape(n, sp.Nx, sp.Nhw, 1, 1)
            if want("ζ_dϕ"): res["ζ_dϕ"] = basis(sp.H, 1)
            if want("ζω"): res["ζω"] = basis(sp.Hw, 0).transpose(1, 2)
            if want("ω"): res["ω"] = dev["ω"].reshape(n, 1, sp.Nc)
            if "rho_real" in dev and want("rho_real"):
                res["rho_real"] = dev["rho_real"].reshape(n, sp.Ng, 1)
        return res

    def sample_posterior(self, num_samples=1, rs=None, mp=None, take_mean=True):
        """`Predictive(model, guide=guide, num_samples=n, return_sites=rs)`: n guide draws pushed through
        the model; returns {site: (n, ...) CPU tensor} with Pyro's site shapes."""
        return {k: v.cpu() for k, v in self._draw_sites(num_samples, rs).items()}

    def _site_exists(self, n):
        sp = self.spec
        return {"ν": True, "ϕxy": True, "Δν": sp.with_delta_nu, "shape_inv": sp.noisemodel == "NegativeBinomial",
                "logγg": sp.kind == "velocity", "logβg": sp.kind == "velocity", "νω": sp.kind == "velocity",
                "rho_real": sp.kind == "velocity" and sp.guide == "lrmn"}[n]

    def _binned_posterior(self):
        """num_samples draws in bins of n_per_bin (the reference's memory bound, velocity_inference_model.py:198-232);
        every bin is copied from the device straight into its rows of the result."""
        nbins = int(np.ceil(self.num_samples / self.n_per_bin))
        out = {}
        for i in range(nbins):
            part = self._draw_sites(self.n_per_bin)
            for k, v in part.items():
                if k not in out:
                    out[k] = torch.empty((nbins * self.n_per_bin,) + tuple(v.shape[1:]), dtype=v.dtype)
                out[k][i * self.n_per_bin:(i + 1) * self.n_per_bin].copy_(v)
        return out

    def check_model(self):
        st = self.engine.stats if self.engine else {}
        print({"kind": self._kind, "spec": None if self.engine is None else
               (self.spec.Ng, self.spec.Nc, self.spec.guide, self.spec.noisemodel), **st})

    check_guide = check_model


class PhaseFitModel(_FitBase):
    _kind = "phase"

    def __init__(self, metaparams, condition_on={}, early_exit=False, get_posterior=True, num_samples=500,
                 n_per_bin=50):
        super().__init__(metaparams, condition_on, early_exit, True, num_samples, n_per_bin)   # reference forces True

    def _posterior(self):
        sp, mp = self.spec, self.metaparams
        post = self._binned_posterior()
        self.metaparams_avg = mp._replace(count_factor=torch.full_like(mp.count_factor, float(mp.count_factor.mean())))
        # ElogS = ν_locs·ζ(ϕ) + Db·Δν_locs + count_factor (ElogS2: with the averaged count factor), on the device
        dnu = pyro_compat.param("Δν_locs") if sp.with_delta_nu else None
        sl = slice(self.engine.c0, self.engine.c1)
        S, S2 = self.engine.expected_logs(pyro_compat.param("ν_locs"), torch.as_tensor(self.phase_pyro.phis)[sl],
                                          float(self.metaparams_avg.count_factor.reshape(-1)[0]), dnu=dnu)
        post["ElogS"], post["ElogS2"] = self._gather(S, 1).squeeze(), self._gather(S2, 1).squeeze()
        self.posterior = post


class VelocityFitModel(_FitBase):
    _kind = "velocity"

    def _warn_sizes(self):
        mp = self.metaparams
        if (mp.Ng < 50) & (mp.Nc < 500):
            print("USER WARNING: the number of genes is below the recommended number for reliable velocity-learning.")
        if (mp.Ng < 350) & (mp.Nc < 50):
            print("USER WARNING: the number of cells is below the recommended number for reliable velocity-learning.")

    def _extract(self):
        par = super()._extract()
        sp, P = self.spec, pyro_compat._STORE
        if sp.guide != "lrmn":
            self.log_gammas = P["logγg_locs"].squeeze().numpy().T
            self.cycle_pyro.set_log_gammas(self.log_gammas)
            self.velocity_coef = P["νω_locs"].unsqueeze(-3).unsqueeze(-4).float().numpy()
            self.velocity_coef_sd = P["νω_scales"].unsqueeze(-3).unsqueeze(-4).float().numpy()
            self.speed_pyro = AngularSpeed.from_array(condition_names=self.metaparams.speed_prior.conditions,
                                                      means_array=self.velocity_coef.squeeze(),
                                                      stds_array=self.velocity_coef_sd.squeeze(), Nhω=sp.Nhw)
        self.log_betas = P["logβg_locs"].squeeze().numpy().T
        self.cycle_pyro.set_log_betas(self.log_betas)
        self.cycle_pyro.set_disp_pyro(getattr(self, "disp_pyro", None))
        return par

    def _posterior(self):
        sp, mp = self.spec, self.metaparams
        post = self._binned_posterior()
        self.metaparams_avg = mp._replace(count_factor=torch.full_like(mp.count_factor, float(mp.count_factor.float().mean())))
        phis = self.phase_pyro.phis
        γg = post["γg"].mean(0).reshape(-1)
        logβg = post["logβg"].mean(0).reshape(-1)
        ζω = torch_basis(phis, der=0, kind="fourier", num_harmonics=sp.Hw).T
        νω = post["νω"].mean(0)
        ω = torch.einsum("...xhgc,hc,xhgc->gc", [νω, ζω, mp.D.cpu().float()]).reshape(-1)     # one speed per cell (N2)
        dnu = pyro_compat.param("Δν_locs") if sp.with_delta_nu else None
        # the four dense (Ng, Nc) summaries of velocity_inference_model.py:236-258 in one device pass
        sl = slice(self.engine.c0, self.engine.c1)
        S, S2, U, U2 = self.engine.expected_logs(pyro_compat.param("ν_locs"), torch.as_tensor(phis)[sl],
                                                 float(self.metaparams_avg.count_factor.reshape(-1)[0]), dnu=dnu,
                                                 omega=ω[sl], logbeta=logβg, gamma=γg)
        post["ElogS"], post["ElogU"] = self._gather(S, 1).squeeze(), self._gather(U, 1).squeeze()
        post["ElogS2"], post["ElogU2"] = self._gather(S2, 1).squeeze(), self._gather(U2, 1).squeeze()
        self.posterior = post
        if sp.guide == "lrmn":                      # velocity_inference_model.py:264-274
            self.log_gammas = post["logγg"].mean(0).squeeze().numpy().T
            self.cycle_pyro.set_log_gammas(self.log_gammas)
            self.velocity_coef = post["νω"].mean(0).float().numpy()
            self.speed_pyro = AngularSpeed.from_array(condition_names=mp.speed_prior.conditions,
                                                      means_array=self.velocity_coef.squeeze(),
                                                      stds_array=post["νω"].std(0).float().squeeze().numpy(),
                                                      Nhω=sp.Nhw)


def run_svi(metaparams, optimizer, num_steps=1000, condition_on={}, kind=None, **fit_kwargs):
    """Thin convenience wrapper named in BASELINE.json's north_star (the reference has no `run_svi`;
    its nearest relative is the notebook-local `fit_SVI`, tutorials/1D_Pancreas_Analysis.ipynb cell 26)."""
    kind = kind or ("velocity" if hasattr(metaparams, "Nx") else "phase")
    cls = VelocityFitModel if kind == "velocity" else PhaseFitModel
    m = cls(metaparams, condition_on=condition_on, get_posterior=fit_kwargs.pop("get_posterior", False))
    m.fit(optimizer, num_steps=num_steps, verbose=False, **fit_kwargs)
    return m
