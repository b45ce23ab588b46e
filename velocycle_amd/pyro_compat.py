"""The few `pyro` module-level names the tutorials use around fit(): a parameter store holding the
fitted variational parameters under Pyro's names and shapes, `clear_param_store()`, `param(name)`."""
from __future__ import annotations

from typing import Dict

import torch

_STORE: Dict[str, torch.Tensor] = {}


def clear_param_store():
    _STORE.clear()


def get_param_store():
    return _STORE


def param(name):
    return _STORE[name]


class Trace_ELBO:
    """Look-alike of `pyro.infer.Trace_ELBO(num_particles=K)` as the tutorials hand it to `fit(loss=...)`
    (velocity_inference_model.py:79,111; phase_inference_model.py:128,162): a holder of `num_particles` -- K guide draws per
    step, loss and gradients averaged over them before the optimiser step -- and of the one piece of state Pyro's object
    carries between fits: a fresh object makes one extra guide pass before its first step (`fresh`, cleared on first use).
    Any object with a `num_particles` attribute (a real pyro ELBO included) is read the same way by fit()."""

    def __init__(self, num_particles: int = 1, **kwargs):
        if int(num_particles) < 1:
            raise ValueError("num_particles must be >= 1")
        unsupported = {k: v for k, v in kwargs.items() if k in ("vectorize_particles",) and v}
        if unsupported:
            raise NotImplementedError(f"Trace_ELBO options not supported by the HIP engine: {sorted(unsupported)}")
        self.num_particles = int(num_particles)
        self.fresh = True


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _clipped_adam(*a, **k):
    from .svi import ClippedAdam
    return ClippedAdam(*a, **k)


# `import velocycle_amd.pyro_compat as pyro` then reads like the tutorials: pyro.infer.Trace_ELBO(...), pyro.optim.ClippedAdam({...})
infer = _Namespace(Trace_ELBO=Trace_ELBO)
optim = _Namespace(ClippedAdam=_clipped_adam)
