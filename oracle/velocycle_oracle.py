"""CPU oracle: a torch restatement of VeloCycle's SVI hot path (ELBO + reparameterised gradient).

TEST INFRASTRUCTURE ONLY.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; nothing under `velocycle_amd/` does.  It is the checker, never the
product: the product path (velocycle_amd -> libvelocycle_hip.so) raises when the HIP library is missing.

What it restates (reference = /root/reference, lamanno-epfl/velocycle v0.1.0.5; line numbers are
for the source tree `velocycle/*.py`):

  * `torch_fourier_basis`          utils.py:400-437      -> fourier_basis()
  * `pack_direction`               utils.py:488-506      -> pack_direction()
  * `phase_latent_variable_model`  phase_inference_model.py:343-395   -> _model_log_joint(kind="phase")
  * `phase_latent_variable_guide`  phase_inference_guide.py:10-56     -> _guide_meanfield()
  * `velocity_latent_variable_model[_LRMN]`  velocity_inference_model.py:304-388 / 390-471
  * `velocity_latent_variable_guide[_LRMN]`  velocity_inference_guide.py:9-63 / 65-141
  * the SVI driver semantics of `PhaseFitModel.fit` / `VelocityFitModel.fit`
    (phase_inference_model.py:162-185, velocity_inference_model.py:111-151)

Third-party arithmetic that is NOT in /root/reference (pyro-ppl==1.8.6, requirements.txt:105,
on torch==2.1.1) is restated from its published algorithm:
  * `Trace_ELBO(num_particles=1)`: loss = -(sum model log-probs - sum guide log-probs), one
    reparameterised sample; conditioned sites are observed in the model (their prior log-prob is a
    constant kept in the loss) and hidden from the guide.
  * `GammaPoisson(c, rate).log_prob(k) = -log_beta(c, k+1) - log(c+k) + c log(rate) - (c+k) log(1+rate)`.
  * `Delta.log_prob = 0` at its own value; `Normal/Gamma/LowRankMultivariateNormal` = torch.distributions.
  * `ClippedAdam` (pyro/optim/clipped_adam.py): lr *= lrd before each update, elementwise clamp of
    the gradient to +-clip_norm, Adam moments, step = lr*sqrt(1-b2^t)/(1-b1^t), denom = sqrt(v)+eps.
  * RNG draw order of one guide call (each draw is `torch.empty(shape).normal_()`): see draw_eps().

PARITY PINNING.  utils.py functions are importable here and pin fourier_basis()/pack_direction()
directly (tests/golden/basis_*.npz).  The model/guide/fit bodies are pinned by executing the
reference's own files UNMODIFIED on top of `oracle/pyro_shim` (tests/golden/make_golden.py ->
tests/golden/ref_*.npz).  Because the shim itself restates Pyro's semantics, parity at the Pyro
boundary is "unpinned": no run of real pyro-ppl is possible in this image.

Everything is written op-by-op the way the reference does it (full (Ng,Nc) temporaries, autograd for
gradients), so in float32 it doubles as the timed CPU baseline ("port") of bench.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

LOG_2PI = math.log(2.0 * math.pi)

# Site names use the reference's exact code points.
PHIXY, NU, DNU, SHAPE_INV = "ϕxy", "ν", "Δν", "shape_inv"
LOGGAMMA, LOGBETA, NUOMEGA, RHO_REAL = "logγg", "logβg", "νω", "rho_real"


# --------------------------------------------------------------------------------------
# helpers restated from utils.py
# --------------------------------------------------------------------------------------
def fourier_basis(phi: torch.Tensor, num_harmonics: int, der: int = 0) -> torch.Tensor:
    """utils.py:400-437.  Columns [1, sin p, cos p, sin 2p, cos 2p, ...] (der=0) or their
    phi-derivative [0, cos p, -sin p, 2cos 2p, -2 sin 2p, ...] (der=1).  (Nc,) -> (Nc, 2H+1)."""
    cols = []
    if der == 0:
        cols.append(torch.ones_like(phi))
        for k in range(1, num_harmonics + 1):
            cols += [torch.sin(k * phi), torch.cos(k * phi)]
    elif der == 1:
        cols.append(torch.zeros_like(phi))
        for k in range(1, num_harmonics + 1):
            cols += [k * torch.cos(k * phi), -k * torch.sin(k * phi)]
    else:
        raise ValueError(f"Value {der=} is not allowed, use 0 or 1 instead")
    return torch.stack(cols, dim=-1)


def pack_direction(xy: torch.Tensor) -> torch.Tensor:
    """utils.py:488-506: atan2(y, x) of the last axis (x = [...,0], y = [...,1])."""
    return torch.atan2(xy[..., 1], xy[..., 0])


def gamma_poisson_log_prob(conc, rate, value):
    """pyro.distributions.GammaPoisson.log_prob (pyro 1.8.6 conjugate.py), op by op."""
    post = conc + value
    log_beta = torch.lgamma(conc) + torch.lgamma(value + 1) - torch.lgamma(conc + value + 1)
    return -log_beta - post.log() + conc * rate.log() - post * (1 + rate).log()


def normal_log_prob(x, loc, scale):
    scale = torch.as_tensor(scale, dtype=x.dtype)
    return -0.5 * ((x - loc) / scale) ** 2 - torch.log(scale) - 0.5 * LOG_2PI


def gamma_log_prob(x, alpha, beta):
    alpha = torch.as_tensor(alpha, dtype=x.dtype)
    beta = torch.as_tensor(beta, dtype=x.dtype)
    return alpha * torch.log(beta) + (alpha - 1) * torch.log(x) - beta * x - torch.lgamma(alpha)


# --------------------------------------------------------------------------------------
# problem container
# --------------------------------------------------------------------------------------
@dataclass
class Problem:
    """All inputs of one fit, in canonical dense shapes (what `MetaparContainer` holds, squeezed)."""
    kind: str                    # "phase" | "velocity"
    guide: str                   # "meanfield" | "lrmn"   (phase is always meanfield)
    noisemodel: str              # "NegativeBinomial" | "Poisson" | "Lognormal"
    with_delta_nu: bool
    H: int                       # harmonics of the expression map
    S: torch.Tensor              # (Ng, Nc)
    count_factor: torch.Tensor   # (Nc,)
    Db: torch.Tensor             # (Nb, Nc)
    mu_nu: torch.Tensor          # (Ng, Nh)
    sd_nu: torch.Tensor          # (Ng, Nh)
    phixy_prior: torch.Tensor    # (Nc, 2)
    U: Optional[torch.Tensor] = None          # (Ng, Nc)   velocity only
    D: Optional[torch.Tensor] = None          # (Nx, Nc)
    Hw: int = 0
    mu_gamma: Optional[torch.Tensor] = None   # (Ng,)
    sd_gamma: Optional[torch.Tensor] = None
    mu_beta: Optional[torch.Tensor] = None
    sd_beta: Optional[torch.Tensor] = None
    mu_nuw: Optional[torch.Tensor] = None     # (Nx, Nhw)
    sd_nuw: Optional[torch.Tensor] = None
    mu_dnu: float = 0.0                       # guide init of Delta-nu
    sd_dnu: object = 0.5                      # phase prior scale: scalar or (Nb, Ng); velocity: 0.01 fixed
    gamma_alpha: float = 1.0
    gamma_beta: float = 2.0
    sigma_ln_s: float = 0.5                   # Lognormal noise scale (phase 0.5; velocity 0.1/0.1)
    sigma_ln_u: float = 0.1
    rho_mean: float = 4.0
    rho_std: float = 1.0
    rho_scale: float = 1.0
    rho_rank: int = 5
    condition_on: Dict[str, torch.Tensor] = field(default_factory=dict)

    @property
    def Ng(self): return self.S.shape[0]
    @property
    def Nc(self): return self.S.shape[1]
    @property
    def Nb(self): return self.Db.shape[0]
    @property
    def Nx(self): return 0 if self.D is None else self.D.shape[0]
    @property
    def Nh(self): return 2 * self.H + 1
    @property
    def Nhw(self): return 2 * self.Hw + 1
    @property
    def dtype(self): return self.S.dtype

    def to(self, dtype):
        kw = {}
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor) and v.is_floating_point():
                kw[k] = v.to(dtype)
            elif k == "condition_on":
                kw[k] = {a: b.to(dtype) for a, b in v.items()}
            else:
                kw[k] = v
        return Problem(**kw)

    def cond(self, name):
        """Conditioned value of a site in canonical shape, or None."""
        if name not in self.condition_on:
            return None
        v = self.condition_on[name].to(self.dtype)
        if name == PHIXY:
            return v.reshape(self.Nc, 2)
        if name == NU:
            return v.reshape(self.Ng, self.Nh)
        if name == DNU:
            return v.reshape(self.Nb, self.Ng)
        if name == NUOMEGA:
            return v.reshape(self.Nx, self.Nhw)
        return v.reshape(self.Ng)


def problem_from_metaparams(mp, kind: str, condition_on=None, dtype=torch.float64) -> Problem:
    """Build a Problem from a reference-style MetaparContainer (preprocessing.py:168-205, 270-323)."""
    condition_on = dict(condition_on or {})
    f = lambda t: torch.as_tensor(t).detach().to(dtype)
    Ng, Nc = int(mp.Ng), int(mp.Nc)
    common = dict(
        kind=kind, noisemodel=mp.noisemodel, with_delta_nu=bool(mp.with_delta_nu),
        S=f(mp.S).reshape(Ng, Nc), count_factor=f(mp.count_factor).reshape(Nc),
        Db=f(mp.Db).reshape(int(mp.Nb), Nc),
        mu_nu=f(mp.μνg).reshape(Ng, -1), sd_nu=f(mp.σνg).reshape(Ng, -1),
        phixy_prior=f(mp.φxy_prior).reshape(Nc, 2),
        mu_dnu=float(mp.μΔν), gamma_alpha=float(mp.gamma_alpha), gamma_beta=float(mp.gamma_beta),
        condition_on={k: f(v) for k, v in condition_on.items()},
    )
    if kind == "phase":
        sd = f(mp.σΔν)
        return Problem(guide="meanfield", H=int(mp.num_harmonics_S),
                       sd_dnu=(float(sd) if sd.numel() == 1 else sd.reshape(int(mp.Nb), Ng)),
                       sigma_ln_s=float(mp.σgc), **common)
    Nx = int(mp.Nx)
    return Problem(
        guide=("lrmn" if mp.model_type == "lrmn" else "meanfield"), H=int(mp.num_harmonics),
        U=f(mp.U).reshape(Ng, Nc), D=f(mp.D).reshape(Nx, Nc), Hw=(int(mp.Nhω) - 1) // 2,
        mu_gamma=f(mp.μγ).reshape(Ng), sd_gamma=f(mp.σγ).reshape(Ng),
        mu_beta=f(mp.μβ).reshape(Ng), sd_beta=f(mp.σβ).reshape(Ng),
        mu_nuw=f(mp.μνω).reshape(Nx, -1), sd_nuw=f(mp.σνω).reshape(Nx, -1),
        sd_dnu=0.01, sigma_ln_s=float(mp.σsgc), sigma_ln_u=float(mp.σugc),
        rho_mean=float(mp.rho_mean), rho_std=float(mp.rho_std), rho_scale=float(mp.rho_scale),
        rho_rank=int(mp.rho_rank), **common)


# --------------------------------------------------------------------------------------
# parameters (the pyro.param store of the guides), unconstrained
# --------------------------------------------------------------------------------------
POSITIVE = {"ν_scales", "logγg_scales", "logβg_scales", "νω_scales", "shape_inv_locs",
            "cov_factor", "cov_diag"}


def init_params(p: Problem, cov_factor_draw: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Initial values of every `pyro.param` of the guide, UNCONSTRAINED (positive ones stored as log),
    in canonical shapes.  phase_inference_guide.py:36-45; velocity_inference_guide.py:25-43 / 78-102.
    For the LRMN guide `cov_factor`'s init is the `torch.normal(zeros, 0.02*ones)` the reference
    evaluates as an argument of `pyro.param` on EVERY guide call (velocity_inference_guide.py:91-92);
    only the first call's draw is kept, so pass `draw_eps(...)["_cov_factor_draw"]` of that call."""
    dt = p.dtype
    out: Dict[str, torch.Tensor] = {}
    out["ν_locs"] = p.mu_nu.clone()
    out["ν_scales"] = p.sd_nu.log()
    if p.with_delta_nu:
        out["Δν_locs"] = torch.full((p.Nb, p.Ng), float(p.mu_dnu), dtype=dt)
    out["ϕxy_locs"] = p.phixy_prior.clone()
    if p.kind == "velocity":
        if p.guide == "meanfield":
            out["logγg_locs"] = p.mu_gamma.clone()
            out["logγg_scales"] = p.sd_gamma.log()
            out["logβg_locs"] = p.mu_beta.clone()
            out["logβg_scales"] = p.sd_beta.log()
            out["νω_locs"] = p.mu_nuw.clone()
            out["νω_scales"] = p.sd_nuw.log()
        else:
            M, R = p.Ng + p.Nx * p.Nhw, p.rho_rank
            out["logβg_locs"] = p.mu_beta.clone()
            out["logβg_scales"] = p.sd_beta.log()
            out["loc"] = torch.cat([p.mu_gamma, p.mu_nuw.reshape(-1)])
            if cov_factor_draw is None:
                raise ValueError("LRMN guide needs the first guide call's cov_factor draw")
            out["cov_factor"] = torch.clip(cov_factor_draw, min=0).to(dt).log()   # log(0) = -inf, as in Pyro
            out["cov_diag"] = (torch.cat([p.sd_gamma, p.sd_nuw.reshape(-1)]) ** 2).log()
            out["rho_real_loc"] = torch.full((p.Ng,), float(p.rho_mean), dtype=dt)
    if p.noisemodel == "NegativeBinomial":
        out["shape_inv_locs"] = torch.full((p.Ng,), math.log(p.gamma_alpha / p.gamma_beta), dtype=dt)
    return out


def constrained(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: (v.exp() if k in POSITIVE else v) for k, v in params.items()}


def draw_eps(p: Problem, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """Standard-normal draws of ONE guide call in the reference's order and shapes.
    Draws happen for conditioned (blocked) sites too.  Always drawn in float32 like the reference,
    then cast to the problem dtype."""
    def n(*shape):
        return torch.empty(shape, dtype=torch.float32).normal_(generator=generator).to(p.dtype)
    e: Dict[str, torch.Tensor] = {}
    if p.kind == "phase":
        e[NU] = n(p.Ng, 1, p.Nh).reshape(p.Ng, p.Nh)
        e[PHIXY] = n(p.Nc, 2)
    elif p.guide == "meanfield":
        e[LOGGAMMA] = n(p.Ng, 1).reshape(p.Ng)
        e[LOGBETA] = n(p.Ng, 1).reshape(p.Ng)
        e[NU] = n(p.Ng, 1, p.Nh).reshape(p.Ng, p.Nh)
        e[NUOMEGA] = n(p.Nx, p.Nhw, 1, 1).reshape(p.Nx, p.Nhw)
        e[PHIXY] = n(p.Nc, 2)
    else:
        M, R = p.Ng + p.Nx * p.Nhw, p.rho_rank
        # `cov_factor`'s init expression: drawn on every call, kept only by the first one
        e["_cov_factor_draw"] = torch.normal(torch.zeros((M, R)), torch.ones((M, R)) * 0.02, generator=generator)
        e["eps_W"] = n(R)
        e["eps_D"] = n(M)
        e[NU] = n(p.Ng, 1, p.Nh).reshape(p.Ng, p.Nh)
        e[LOGBETA] = n(p.Ng, 1).reshape(p.Ng)
        e[PHIXY] = n(p.Nc, 2)
    return e


# --------------------------------------------------------------------------------------
# guide: samples + log q
# --------------------------------------------------------------------------------------
def _guide(p: Problem, par: Dict[str, torch.Tensor], eps: Dict[str, torch.Tensor]):
    """Runs the guide with explicit eps.  Returns (values of every guide site, log q summed over the
    sites that are NOT hidden by conditioning).  Delta sites contribute 0."""
    c = constrained(par)
    hid = set(p.condition_on)
    val: Dict[str, torch.Tensor] = {}
    logq = torch.zeros((), dtype=p.dtype)

    def normal_site(name, loc, scale, e):
        nonlocal logq
        x = loc + scale * e
        val[name] = x
        if name not in hid:
            logq = logq + (-0.5 * e ** 2 - torch.log(torch.as_tensor(scale, dtype=p.dtype)) - 0.5 * LOG_2PI
                           + torch.zeros_like(x)).sum()

    if p.kind == "velocity" and p.guide == "lrmn":
        # velocity_inference_guide.py:89-139
        W, Dg, loc = c["cov_factor"], c["cov_diag"], c["loc"]
        X = loc + W @ eps["eps_W"] + Dg.sqrt() * eps["eps_D"]
        lg = X[: p.Ng]
        val[LOGGAMMA] = lg
        normal_site(NU, c["ν_locs"], c["ν_scales"], eps[NU])
        val[RHO_REAL] = c["rho_real_loc"]
        rho = torch.sigmoid(c["rho_real_loc"] / p.rho_scale) * 1.998 - 0.999
        s_gamma = torch.sqrt((W @ W.T + torch.diag(Dg)).diagonal()[: p.Ng])
        mu_b = c["logβg_locs"] + rho * c["logβg_scales"] * (lg - loc[: p.Ng]) / s_gamma
        sd_b = c["logβg_scales"] * torch.sqrt(1 - rho ** 2)
        normal_site(LOGBETA, mu_b, sd_b, eps[LOGBETA])
        val[NUOMEGA] = X[p.Ng:].reshape(p.Nx, p.Nhw)
    elif p.kind == "velocity":
        # velocity_inference_guide.py:45-60
        normal_site(LOGGAMMA, c["logγg_locs"], c["logγg_scales"], eps[LOGGAMMA])
        normal_site(LOGBETA, c["logβg_locs"], c["logβg_scales"], eps[LOGBETA])
        normal_site(NU, c["ν_locs"], c["ν_scales"], eps[NU])
        normal_site(NUOMEGA, c["νω_locs"], c["νω_scales"], eps[NUOMEGA])
    else:
        normal_site(NU, c["ν_locs"], c["ν_scales"], eps[NU])
    if p.with_delta_nu:
        val[DNU] = c["Δν_locs"]
    if p.noisemodel == "NegativeBinomial":
        val[SHAPE_INV] = c["shape_inv_locs"]
    normal_site(PHIXY, c["ϕxy_locs"], 1.0, eps[PHIXY])
    return val, logq


# --------------------------------------------------------------------------------------
# model: log joint at given site values
# --------------------------------------------------------------------------------------
def _model_log_joint(p: Problem, v: Dict[str, torch.Tensor]):
    """log p(data, sites) following phase_inference_model.py:360-395 /
    velocity_inference_model.py:322-388 (and :404-471).  Returns (log_joint, deterministic sites)."""
    lp = torch.zeros((), dtype=p.dtype)
    det: Dict[str, torch.Tensor] = {}
    nu = v[NU]
    lp = lp + normal_log_prob(nu, p.mu_nu, p.sd_nu).sum()
    if p.with_delta_nu:
        dnu = v[DNU]
        lp = lp + normal_log_prob(dnu, 0.0, p.sd_dnu if p.kind == "phase" else 0.01).sum()
    xy = v[PHIXY]
    lp = lp + normal_log_prob(xy, p.phixy_prior, 1.0).sum()
    phi = pack_direction(xy)
    zeta = fourier_basis(phi, p.H, der=0)                       # (Nc, Nh)
    ElogS = torch.einsum("gh,ch->gc", nu, zeta) + p.count_factor
    if p.with_delta_nu:
        ElogS = ElogS + torch.einsum("bc,bg->gc", p.Db, dnu)
    det.update({"ϕ": phi, "ζ": zeta, "ElogS": ElogS})

    if p.kind == "velocity":
        lg, lb, nuw = v[LOGGAMMA], v[LOGBETA], v[NUOMEGA]
        lp = lp + normal_log_prob(lg, p.mu_gamma, p.sd_gamma).sum()
        lp = lp + normal_log_prob(lb, p.mu_beta, p.sd_beta).sum()
        lp = lp + normal_log_prob(nuw, p.mu_nuw, p.sd_nuw).sum()
        if p.guide == "lrmn":
            lp = lp + normal_log_prob(v[RHO_REAL], p.rho_mean, p.rho_std).sum()
        gam = torch.exp(lg)
        zeta_d = fourier_basis(phi, p.H, der=1)
        zeta_w = fourier_basis(phi, p.Hw, der=0)                # (Nc, Nhw)
        omega = torch.einsum("xh,ch,xc->c", nuw, zeta_w, p.D)   # one speed per cell
        z = torch.einsum("gh,ch->gc", nu, zeta_d) * omega + gam[:, None]
        ElogU = -lb[:, None] + torch.log(torch.relu(z) + 1e-5) + ElogS
        det.update({"γg": gam, "ζ_dϕ": zeta_d, "ζω": zeta_w.T, "ω": omega, "ElogU": ElogU})

    nm = p.noisemodel
    if nm == "NegativeBinomial":
        si = v[SHAPE_INV]
        lp = lp + gamma_log_prob(si, p.gamma_alpha, p.gamma_beta).sum()
        conc = (1.0 / si)[:, None]
        lp = lp + gamma_poisson_log_prob(conc, 1.0 / (si[:, None] * torch.exp(ElogS)), p.S).sum()
        if p.kind == "velocity":
            lp = lp + gamma_poisson_log_prob(conc, 1.0 / (si[:, None] * torch.exp(ElogU)), p.U).sum()
    elif nm == "Poisson":
        lp = lp + (p.S * ElogS - torch.exp(ElogS) - torch.lgamma(p.S + 1)).sum()
        if p.kind == "velocity":
            lp = lp + (p.U * ElogU - torch.exp(ElogU) - torch.lgamma(p.U + 1)).sum()
    elif nm == "Lognormal":
        logS = torch.log(p.S.double() + 1 + 1e-16).float().to(p.dtype)     # preprocessing.py:154,267
        lp = lp + normal_log_prob(logS, ElogS, p.sigma_ln_s).sum()
        if p.kind == "velocity":
            logU = torch.log(p.U.double() + 1 + 1e-16).float().to(p.dtype)
            lp = lp + normal_log_prob(logU, ElogU, p.sigma_ln_u).sum()
    else:
        raise ValueError(f"{nm} not allowed")
    return lp, det


def elbo_loss(p: Problem, par: Dict[str, torch.Tensor], eps: Dict[str, torch.Tensor]):
    """-ELBO of Trace_ELBO(num_particles=1) for explicit eps.  Returns (loss, site values, det sites)."""
    gval, logq = _guide(p, par, eps)
    val = dict(gval)
    for name in p.condition_on:
        val[name] = p.cond(name)
    logp, det = _model_log_joint(p, val)
    return -(logp - logq), val, det


class _NoPath(torch.Tensor):
    """A zero gradient that records that autograd found NO path from the loss to the parameter tensor (`.grad is None` in Pyro:
    PyroOptim skips such a tensor altogether -- it takes no weight decay either)."""


def loss_and_grads(p: Problem, par: Dict[str, torch.Tensor], eps: Dict[str, torch.Tensor]):
    """(loss float, {param name: d loss / d unconstrained param}); params untouched by the loss get 0 (marked `_NoPath`)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in par.items()}
    loss, val, det = elbo_loss(p, leaves, eps)
    loss.backward()
    grads = {k: (torch.zeros_like(v).as_subclass(_NoPath) if v.grad is None else torch.nan_to_num(v.grad, nan=0.0))
             for k, v in leaves.items()}
    return loss.item(), grads, {k: t.detach() for k, t in val.items()}, {k: t.detach() for k, t in det.items()}


# --------------------------------------------------------------------------------------
# optimiser + fit loop
# --------------------------------------------------------------------------------------
class ClippedAdam:
    """pyro.optim.ClippedAdam restated for a dict of tensors (one state per tensor, as PyroOptim)."""

    def __init__(self, optim_args: dict):
        a = dict(optim_args)
        self.lr = a.get("lr", 1e-3)
        self.betas = tuple(a.get("betas", (0.9, 0.999)))
        self.eps = a.get("eps", 1e-8)
        self.clip_norm = a.get("clip_norm", 10.0)
        self.lrd = a.get("lrd", 1.0)
        self.weight_decay = a.get("weight_decay", 0.0)
        self.t = 0
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}

    def step(self, par: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]):
        self.lr *= self.lrd
        self.t += 1
        b1, b2 = self.betas
        step_size = self.lr * math.sqrt(1 - b2 ** self.t) / (1 - b1 ** self.t)
        for k, g in grads.items():
            if isinstance(g, _NoPath):                       # pyro: `if p.grad is None: continue`
                continue
            g = g.clamp(-self.clip_norm, self.clip_norm)
            if self.weight_decay != 0:                       # pyro clipped_adam.py: grad.add(p.data, alpha=weight_decay), behind the clamp
                g = g.add(par[k], alpha=self.weight_decay)
            m = self.m.setdefault(k, torch.zeros_like(g))
            v = self.v.setdefault(k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            upd = m / (v.sqrt() + self.eps)
            par[k] = par[k] - step_size * torch.nan_to_num(upd, nan=0.0)
        return par


class Adam:
    """pyro.optim.Adam = torch.optim.Adam restated for a dict of tensors (tutorials/1D_Pancreas_Analysis.ipynb cell 26 passes it to
    the same SVI loop): no clamp, no decay, weight decay in front of the moments, eps INSIDE the second bias correction."""

    def __init__(self, optim_args: dict):
        a = dict(optim_args)
        self.lr = a.get("lr", 1e-3)
        self.betas = tuple(a.get("betas", (0.9, 0.999)))
        self.eps = a.get("eps", 1e-8)
        self.weight_decay = a.get("weight_decay", 0.0)
        self.t = 0
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}

    def step(self, par: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor]):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k, g in grads.items():
            if isinstance(g, _NoPath):                       # torch.optim.Adam: `if p.grad is not None`
                continue
            if self.weight_decay != 0:
                g = g.add(par[k], alpha=self.weight_decay)
            m = self.m.setdefault(k, torch.zeros_like(g))
            v = self.v.setdefault(k, torch.zeros_like(g))
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            upd = m / (v.sqrt() / math.sqrt(bc2) + self.eps)
            par[k] = par[k] - (self.lr / bc1) * torch.nan_to_num(upd, nan=0.0)
        return par


def fit(p: Problem, optim_args: dict, num_steps: int, seed: Optional[int] = None,
        eps_list=None, params=None, warmup_draw: bool = True, num_particles: int = 1, opt=None):
    """The SVI loop of `XFitModel.fit` with verbose=False, early_exit=False.  With `seed`, eps is
    drawn from a torch CPU generator in the reference's order, including (warmup_draw) the extra
    guide pass Trace_ELBO makes before its first step.  num_particles = K: `Trace_ELBO(num_particles=K)` -- K guide
    draws per step, one after the other, loss and gradients averaged over them before the optimiser step (the `loss=`
    argument of fit(), velocity_inference_model.py:79,111); eps_list then holds num_steps * K draws.
    `params` + `opt` (a ClippedAdam that has stepped before) continue an earlier fit the way a second `fit()` does when the
    param store was not cleared: `pyro.param(name, init)` returns the stored value (velocity_inference_guide.py:25-43,
    phase_inference_guide.py:36-45) and PyroOptim keeps one optimiser state per parameter tensor, so the SAME optimizer object
    carries its moments, step count and decayed learning rate on; a new optimizer object starts them afresh.
    Returns (losses, final unconstrained params)."""
    gen = None
    if seed is not None:
        gen = torch.Generator().manual_seed(seed)
    opt = ClippedAdam(optim_args) if opt is None else opt
    losses = []
    first = draw_eps(p, gen) if (eps_list is None and warmup_draw) else None
    K = int(num_particles)
    for i in range(num_steps):
        loss, grads = 0.0, None
        for k in range(K):
            eps = eps_list[i * K + k] if eps_list is not None else draw_eps(p, gen)
            if params is None:
                src = first if first is not None else eps
                params = init_params(p, src.get("_cov_factor_draw"))
            lk, gk, _, _ = loss_and_grads(p, params, eps)
            loss += lk / K
            grads = {n: g / K for n, g in gk.items()} if grads is None else {n: grads[n] + g / K for n, g in gk.items()}
        losses.append(loss)
        params = opt.step(params, grads)
    return losses, params
