"""Synthetic benchmark workloads: data from `simulate_counts` (recipe of reference
utils.py:508-584) with priors set the way the tutorials set them (Tutorial_Capolupo cells 19-21, 38-41):
cycle prior means [log mean S, 0, 0], stds [std(log(S+1))/2, x0.5, x0.5]; phase prior = noisy true phase
on a circle of radius `concentration`; AngularSpeed.trivial_prior (mean 0, std 3 on the constant term,
0.05 on the harmonics); count_factor = log(colsum / mean colsum) (preprocessing.py:149-152)."""
from __future__ import annotations

import math

import torch

from .simulate import simulate_counts
from .spec import ModelSpec


def make_velocity_spec(Nc=10000, Ng=500, mode="vjoint", n_conditions=1, Hw=1, seed=0, device="cpu",
                       noisemodel="NegativeBinomial", concentration=5.0, sim=None, n_batches=None, H=1) -> ModelSpec:
    """mode: "vjoint" (mean-field guide, nothing conditioned), "vcond" (default LRMN guide conditioned
    on ϕxy, ν, shape_inv [, Δν] like the tutorials), "vcond_mf" (same conditioning, mean-field), "vjoint_lrmn" (LRMN guide,
    nothing conditioned: the reference's default `model_type` without `condition_on`).
    `sim`: a stored output of `simulate_counts` (the sampling kernels of torch are not bit-reproducible across hosts, so
    fixtures that must describe the SAME data on every machine carry the simulated counts: tests/golden/oracle_fit_data_*)."""
    # n_batches (default: = n_conditions, the tutorials' layout): samples of Nc cells each; with more batches than conditions the last
    # condition takes the remaining samples (batch design Db and condition design D are separate arguments of
    # preprocess_for_velocity_estimation, preprocessing.py:207-240)
    n_samples = n_conditions if n_batches is None else int(n_batches)
    assert n_samples >= n_conditions
    omegas = (0.4, 0.3, 0.35, 0.25, 0.45, 0.2, 0.38, 0.28)[:n_samples]
    if sim is None:
        sim = simulate_counts(Nc, Ng, omegas=omegas, seed=seed, device=device)
    S_cm, U_cm = sim["S"], sim["U"]                  # (Nc_total, Ng) cell-major, like AnnData layers
    nct = S_cm.shape[0]
    colsum = S_cm.sum(1)
    cf = torch.log(colsum / colsum.mean()).cpu()
    meanS = S_cm.mean(0).clamp_min(1e-3)
    nu0 = torch.log(meanS).cpu()
    nu0std = (torch.log(S_cm + 1).std(0) / 2).clamp_min(0.05).cpu()
    # H harmonics of the expression map (preprocess_for_* default n_harmonics=2, preprocessing.py:108,217; the tutorials pass 1): the
    # simulated data has one, the higher ones get the same zero-mean prior
    mu_nu = torch.stack([nu0] + [torch.zeros_like(nu0)] * (2 * H), 1)
    sd_nu = torch.stack([nu0std] + [0.5 * nu0std] * (2 * H), 1)
    g = torch.Generator().manual_seed(seed + 7)
    phi0 = sim["phis"].cpu() + 0.3 * torch.randn(nct, generator=g)
    pxy = concentration * torch.stack([torch.cos(phi0), torch.sin(phi0)], 1)
    batch = sim["batch"]
    Db = torch.stack([(batch == b).float() for b in range(n_samples)])         # (Nb, Nc)
    D = torch.stack([((batch == b) if b < n_conditions - 1 else (batch >= b)).float() for b in range(n_conditions)])       # (Nx, Nc)
    with_dnu = n_samples > 1
    Nhw = 2 * Hw + 1
    mu_w = torch.zeros(n_conditions, Nhw)
    sd_w = torch.full((n_conditions, Nhw), 0.05)
    sd_w[:, 0] = 3.0
    spec = ModelSpec(
        kind="velocity", guide="meanfield" if mode in ("vjoint", "vcond_mf") else "lrmn",
        noisemodel=noisemodel, with_delta_nu=with_dnu, H=H, Hw=Hw,
        S=S_cm.t(), U=U_cm.t(), count_factor=cf, Db=Db, D=D,
        mu_nu=mu_nu, sd_nu=sd_nu, phixy_prior=pxy,
        mu_gamma=torch.zeros(Ng), sd_gamma=torch.full((Ng,), 0.5),
        mu_beta=torch.full((Ng,), 2.0), sd_beta=torch.full((Ng,), 3.0),
        mu_nuw=mu_w, sd_nuw=sd_w, sd_dnu=0.01, sigma_ln_s=0.1, sigma_ln_u=0.1)
    if mode.startswith("vcond"):
        true_nu = sim["nu"].cpu()
        if H > 1:
            true_nu = torch.cat([true_nu, torch.zeros(true_nu.shape[0], 2 * (H - 1))], 1)
        spec.condition_on = {"ϕxy": torch.stack([torch.cos(sim["phis"].cpu()), torch.sin(sim["phis"].cpu())], 1),
                             "ν": true_nu, "shape_inv": sim["shape_inv"].cpu()}
        if with_dnu:
            spec.condition_on["Δν"] = torch.zeros(n_samples, Ng)
    spec.truth = sim
    return spec


def make_phase_spec(Nc=3000, Ng=200, seed=0, device="cpu", noisemodel="NegativeBinomial", sim=None, n_batches=1, H=1) -> ModelSpec:
    """n_batches > 1: that many samples of Nc cells each with a one-hot batch design and per-batch offsets (with_delta_nu)."""
    v = make_velocity_spec(Nc, Ng, "vjoint", n_batches, 0, seed, device, noisemodel, sim=sim, H=H)
    spec = ModelSpec(kind="phase", guide="meanfield", noisemodel=noisemodel, with_delta_nu=n_batches > 1, H=H,
                     S=v.S, count_factor=v.count_factor, Db=v.Db, mu_nu=v.mu_nu, sd_nu=v.sd_nu,
                     phixy_prior=v.phixy_prior, sigma_ln_s=0.5)
    spec.truth = v.truth
    return spec
