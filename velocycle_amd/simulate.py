"""Synthetic spliced/unspliced count matrices following the generative recipe of the reference's
`utils.simulate_data` (reference velocycle/utils.py:508-584), vectorised and device-agnostic.

Recipe (utils.py:509-543): per gene (nu0, nu1sin, nu1cos, log gamma, log beta) ~ MVN(mean
[0.4,0,0,0,2.0], std [1.2,0.2,0.2,0.5,1.0], correlation matrix as at :510-517); shape_inv ~ Gamma(1,2);
phi ~ U(0, 2pi); ElogS = nu . zeta(phi); ElogU = -log beta + log(relu(nu . zeta'(phi) * omega + gamma) + 1e-5)
+ ElogS; S, U ~ GammaPoisson(1/shape_inv, 1/(shape_inv * exp(E))).  One block of `Nc` cells per entry of
`omegas` (the reference's `omegas_to_test`), sharing the gene parameters and the phases.
"""
from __future__ import annotations

import math
import numpy as np
import torch

_MEANS = [0.4, 0.0, 0.0, 0.0, 2.0]
_STDS = [1.2, 0.2, 0.2, 0.5, 1.0]
_CORR = [[1.0, 0.05, 0.05, 0.05, 0.30],
         [0.05, 1.0, 0.0, 0.0, 0.0],
         [0.05, 0.0, 1.0, 0.0, 0.0],
         [0.05, 0.0, 0.0, 1.0, 0.30],
         [0.30, 0.0, 0.0, 0.30, 1.0]]


def simulate_counts(Nc=5000, Ng=500, omegas=(0.4,), seed=0, device="cpu", chunk=8192):
    """Returns a dict: S, U (float32, (Nc_total, Ng) cell-major like AnnData layers), `batch`
    (int64 (Nc_total,)), and the ground truth (phis, nu (Ng,3), log_gamma, log_beta, shape_inv, omega)."""
    dev = torch.device(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = torch.tensor(_STDS, dtype=torch.float64)
    cov = torch.diag(sd) @ torch.tensor(_CORR, dtype=torch.float64) @ torch.diag(sd)
    L = torch.linalg.cholesky(cov)
    z = torch.randn(Ng, 5, generator=g, dtype=torch.float64)
    par = (torch.tensor(_MEANS, dtype=torch.float64) + z @ L.T).float()
    nu, lg, lb = par[:, :3].to(dev), par[:, 3].to(dev), par[:, 4].to(dev)
    shape_inv = torch.distributions.Gamma(1.0, 2.0).sample((Ng,)) if False else \
        (-torch.log(torch.rand(Ng, generator=g, dtype=torch.float64)) / 2.0).float()   # Gamma(1,2) = Exp(2)
    shape_inv = shape_inv.clamp_min(1e-3).to(dev)
    phis = (torch.rand(Nc, generator=g, dtype=torch.float64) * 2 * math.pi).float().to(dev)
    gd = torch.Generator(device=dev).manual_seed(seed + 1)
    S_blocks, U_blocks, batch = [], [], []
    r = 1.0 / shape_inv
    for b, om in enumerate(omegas):
        for c0 in range(0, Nc, chunk):
            ph = phis[c0:c0 + chunk]
            zeta = torch.stack([torch.ones_like(ph), torch.sin(ph), torch.cos(ph)], -1)      # (c,3)
            zeta_d = torch.stack([torch.zeros_like(ph), torch.cos(ph), -torch.sin(ph)], -1)
            ElogS = zeta @ nu.T                                                               # (c,Ng)
            ElogU = -lb + torch.log(torch.relu((zeta_d @ nu.T) * om + torch.exp(lg)) + 1e-5) + ElogS
            for E, out in ((ElogS, S_blocks), (ElogU, U_blocks)):
                rate = torch._standard_gamma(r.expand_as(E).contiguous(), generator=gd) / (r / torch.exp(E))
                out.append(torch.poisson(rate.clamp_max(1e6), generator=gd))
        batch.append(torch.full((Nc,), b, dtype=torch.int64))
    return dict(S=torch.cat(S_blocks), U=torch.cat(U_blocks), batch=torch.cat(batch),
                phis=phis.repeat(len(omegas)), nu=nu, log_gamma=lg, log_beta=lb, shape_inv=shape_inv,
                omegas=tuple(omegas))
