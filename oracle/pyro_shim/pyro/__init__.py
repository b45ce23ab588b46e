"""Minimal stand-in for the `pyro` API surface that VeloCycle's model/guide/fit code touches.

TEST INFRASTRUCTURE ONLY.  pyro-ppl==1.8.6 (reference requirements.txt:105) is not installed in the
build container and cannot be installed, so the reference's own
`build/lib/velocycle/{phase,velocity}_inference_{model,guide}.py` cannot be imported as they are.
This package is put on `sys.path` *only* by `tests/golden/make_golden.py` (and the optional
reference cross-check tests) so that those reference files can be executed UNMODIFIED, from where
they lie under /root/reference, to produce golden vectors.  Nothing under `velocycle_amd/` imports it.

What is real and what is restated:
  * the model / guide / fit-driver bodies that run on top of this shim are the reference's own code;
  * the semantics of `pyro.sample / param / plate / deterministic`, `poutine.condition / block`,
    `Trace_ELBO`, `SVI.step`, `Predictive` and `ClippedAdam` are RESTATED here from the published
    behaviour of pyro-ppl 1.8.6 -> parity at the Pyro boundary stays "unpinned" (see DESIGN.md).
"""
from . import runtime as _rt
from .runtime import (sample, param, plate, deterministic, clear_param_store, get_param_store,
                      set_rng_seed)
from . import distributions, poutine, infer, optim  # noqa: F401

__version__ = "1.8.6+shim"
