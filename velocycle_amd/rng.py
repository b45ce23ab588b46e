"""Host-side eps stream that consumes a torch CPU generator exactly like one call of the reference's
guide does: every reparameterised Normal site draws `torch.empty(shape).normal_()`
in program order -- also for sites hidden by `poutine.block` -- and the LRMN guide additionally
evaluates `torch.normal(zeros, 0.02)` (its `cov_factor` init expression,
velocity_inference_guide.py:91-92) and `LowRankMultivariateNormal.rsample` (eps_W, then eps_D) first.
Used for seed-for-seed parity with the reference; the performance path draws eps on the GPU (Philox)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .spec import ModelSpec


def draw_eps(sp: ModelSpec, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    def n(*shape):
        return torch.empty(shape, dtype=torch.float32).normal_(generator=generator)
    e: Dict[str, torch.Tensor] = {}
    if sp.kind == "phase":                      # phase_inference_guide.py:47-56
        e["ν"] = n(sp.Ng, 1, sp.Nh).reshape(sp.Ng, sp.Nh)
        e["ϕxy"] = n(sp.Nc, 2)
    elif sp.guide == "meanfield":               # velocity_inference_guide.py:45-63
        e["logγg"] = n(sp.Ng, 1).reshape(sp.Ng)
        e["logβg"] = n(sp.Ng, 1).reshape(sp.Ng)
        e["ν"] = n(sp.Ng, 1, sp.Nh).reshape(sp.Ng, sp.Nh)
        e["νω"] = n(sp.Nx, sp.Nhw, 1, 1).reshape(sp.Nx, sp.Nhw)
        e["ϕxy"] = n(sp.Nc, 2)
    else:                                       # velocity_inference_guide.py:89-141
        M, R = sp.Ng + sp.Nx * sp.Nhw, sp.rho_rank
        e["_cov_factor_draw"] = torch.normal(torch.zeros((M, R)), torch.ones((M, R)) * 0.02, generator=generator)
        e["eps_W"] = n(R)
        e["eps_D"] = n(M)
        e["ν"] = n(sp.Ng, 1, sp.Nh).reshape(sp.Ng, sp.Nh)
        e["logβg"] = n(sp.Ng, 1).reshape(sp.Ng)
        e["ϕxy"] = n(sp.Nc, 2)
    return e
