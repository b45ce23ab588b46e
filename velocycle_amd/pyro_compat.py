"""The few `pyro` module-level names the tutorials use around fit(): a parameter store holding the
fitted variational parameters under Pyro's names and shapes, `clear_param_store()`, `param(name)`."""
from __future__ import annotations

from typing import Dict

import torch

_STORE: Dict[str, torch.Tensor] = {}
# What Pyro's store really holds is the UNCONSTRAINED tensor (the constrained value is a transform of it): the values a
# later fit() continues from, exactly as the last fit left them (canonical engine shape, CPU), next to the published object
# they belong to -- if the user has replaced `_STORE[name]` since, the raw copy no longer describes it and is not used.
_RAW: Dict[str, dict] = {}
# clear_param_store() starts a new generation of parameter tensors: optimiser state kept on an optimizer object refers to
# the tensors of the generation it was made in (PyroOptim keys its per-parameter optimisers by the tensor)
_GENERATION = [0]


# perf-mode (Philox) SVI steps drawn by fit() calls since the store was last cleared: a fit() that CONTINUES from the store but
# whose step counter starts at 0 again (a new optimizer object, a dict, mixed step counts) must not replay the noise of the
# fit it continues -- Pyro draws from a global RNG that keeps advancing (fit_models._perf_seed)
_PERF_DRAWS = [0]


def clear_param_store():
    """`pyro.clear_param_store()`: the next fit() starts from the guides' initial values again, and optimizer objects that
    stepped on the old parameters start afresh on the new ones."""
    _STORE.clear()
    _RAW.clear()
    _GENERATION[0] += 1
    _PERF_DRAWS[0] = 0


def perf_steps_drawn() -> int:
    return _PERF_DRAWS[0]


def add_perf_steps(n: int):
    _PERF_DRAWS[0] += int(n)


def get_param_store():
    return _STORE


def param(name):
    return _STORE[name]


def generation() -> int:
    return _GENERATION[0]


def publish(name: str, constrained: torch.Tensor, unconstrained: torch.Tensor):
    """Called by fit(): the fitted parameter under Pyro's name and shape, and the raw values a later fit() continues from."""
    _STORE[name] = constrained
    _RAW[name] = {"u": unconstrained, "pub": constrained}


def stored_unconstrained(name: str, positive: bool):
    """The unconstrained values of a stored parameter (flat), or None when the store has no such name: the raw copy of the
    last fit when `_STORE[name]` is still the object that fit published, else derived from whatever the user put there."""
    if name not in _STORE:
        return None
    raw = _RAW.get(name)
    if raw is not None and raw["pub"] is _STORE[name]:
        return raw["u"].reshape(-1)
    v = torch.as_tensor(_STORE[name]).detach().float().cpu().reshape(-1)
    return v.log() if positive else v


class Trace_ELBO:
    """Look-alike of `pyro.infer.Trace_ELBO(num_particles=K)` as the tutorials hand it to `fit(loss=...)`
    (velocity_inference_model.py:79,111; phase_inference_model.py:128,162): a holder of `num_particles` -- K guide draws per
    step, loss and gradients averaged over them before the optimiser step -- and of the one piece of state Pyro's object
    carries between fits: a fresh object makes one extra guide pass before its first step (`fresh`, cleared on first use).
    Any object with a `num_particles` attribute (a real pyro ELBO included) is read the same way by fit().

    `vectorize_particles=True` (pyro-ppl 1.8.6 infer/elbo.py: the K particles as one trace under an extra plate) is the same
    estimator -- K draws per step, loss and gradients averaged -- and is what the engine's batched particle step computes anyway
    (vc_svi_run_particles: K_pre / K_post of all particles in one launch each): fit(mode="perf") accepts it.  What differs in
    Pyro is only the ORDER in which the host RNG is consumed (site-major instead of particle-major), which matters to
    mode="parity" alone -- refused there by name, since no fixture pins that order (the reference's own einsum
    "...gch,...ch->gc" sums over a leading particle dimension, so its model cannot run vectorised as it is)."""

    def __init__(self, num_particles: int = 1, max_plate_nesting=float("inf"), vectorize_particles: bool = False, **kwargs):
        if int(num_particles) < 1:
            raise ValueError("num_particles must be >= 1")
        unsupported = {k: v for k, v in kwargs.items() if k not in ("max_iarange_nesting", "strict_enumeration_warning", "ignore_jit_warnings",
                                                                    "jit_options", "retain_graph", "tail_adaptive_beta")}
        if unsupported:
            raise TypeError(f"Trace_ELBO: unexpected arguments {sorted(unsupported)}")
        self.num_particles = int(num_particles)
        self.max_plate_nesting = max_plate_nesting
        self.vectorize_particles = bool(vectorize_particles)
        self.fresh = True


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _clipped_adam(*a, **k):
    from .svi import ClippedAdam
    return ClippedAdam(*a, **k)


def _adam(*a, **k):
    from .svi import Adam
    return Adam(*a, **k)


# `import velocycle_amd.pyro_compat as pyro` then reads like the tutorials: pyro.infer.Trace_ELBO(...), pyro.optim.ClippedAdam({...})
infer = _Namespace(Trace_ELBO=Trace_ELBO)
optim = _Namespace(ClippedAdam=_clipped_adam, Adam=_adam)
