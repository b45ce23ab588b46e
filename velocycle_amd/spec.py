"""`ModelSpec`: the dense, canonical-shape view of a reference `MetaparContainer`
(reference velocycle/preprocessing.py:168-205 and 270-323) that the HIP engine consumes."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch


@dataclass
class ModelSpec:
    kind: str                    # "phase" | "velocity"
    guide: str                   # "meanfield" | "lrmn"
    noisemodel: str              # "NegativeBinomial" | "Poisson" | "Lognormal"
    with_delta_nu: bool
    H: int
    S: Optional[torch.Tensor]    # (Ng, Nc) float32, any strides, CPU or GPU (may be None when S_csr is given)
    count_factor: torch.Tensor   # (Nc,)
    Db: torch.Tensor             # (Nb, Nc)
    mu_nu: torch.Tensor          # (Ng, Nh)
    sd_nu: torch.Tensor
    phixy_prior: torch.Tensor    # (Nc, 2)
    U: Optional[torch.Tensor] = None
    D: Optional[torch.Tensor] = None          # (Nx, Nc)
    Hw: int = 0
    mu_gamma: Optional[torch.Tensor] = None   # (Ng,)
    sd_gamma: Optional[torch.Tensor] = None
    mu_beta: Optional[torch.Tensor] = None
    sd_beta: Optional[torch.Tensor] = None
    mu_nuw: Optional[torch.Tensor] = None     # (Nx, Nhw)
    sd_nuw: Optional[torch.Tensor] = None
    mu_dnu: float = 0.0
    sd_dnu: object = 0.5                      # phase: scalar or (Nb, Ng); velocity fixes 0.01
    gamma_alpha: float = 1.0
    gamma_beta: float = 2.0
    sigma_ln_s: float = 0.5
    sigma_ln_u: float = 0.1
    rho_mean: float = 4.0
    rho_std: float = 1.0
    rho_scale: float = 1.0
    rho_rank: int = 5
    condition_on: Dict[str, torch.Tensor] = field(default_factory=dict)
    # the same count matrices as scipy CSR (cells x genes), e.g. AnnData layers as read from an .h5ad: when present the
    # engine ingests these (vc_set_counts_csr) and never forms / uploads the dense matrices
    S_csr: object = None
    U_csr: object = None

    @property
    def Ng(self): return int(self.S.shape[0]) if self.S is not None else int(self.S_csr.shape[1])
    @property
    def Nc(self): return int(self.S.shape[1]) if self.S is not None else int(self.S_csr.shape[0])
    @property
    def Nb(self): return int(self.Db.shape[0])
    @property
    def Nx(self): return 0 if self.D is None else int(self.D.shape[0])
    @property
    def Nh(self): return 2 * self.H + 1
    @property
    def Nhw(self): return 2 * self.Hw + 1

    def site_shape(self, name):
        return {"ϕxy": (self.Nc, 2), "ν": (self.Ng, self.Nh), "Δν": (self.Nb, self.Ng),
                "νω": (self.Nx, self.Nhw)}.get(name, (self.Ng,))


def spec_from_metaparams(mp, kind: str, condition_on=None) -> ModelSpec:
    """Squeeze a reference-style MetaparContainer into canonical shapes (float32, no copies of S/U)."""
    condition_on = dict(condition_on or {})
    f = lambda t: torch.as_tensor(t).detach().float()
    Ng, Nc = int(mp.Ng), int(mp.Nc)
    from .preprocessing import csr_is_current, raw_field
    # The CSR side channel is used only while it provably describes the same data as the dense field the reference's contract
    # exposes (and users edit with `_replace` / in place): preprocessing tags it with that field.  While it is current the
    # dense field -- a lazily built view of the sparse layer -- is not even touched.
    vel = kind == "velocity"
    S_csr, U_csr = getattr(mp, "S_csr", None), getattr(mp, "U_csr", None) if vel else None
    use_csr = (S_csr is not None and csr_is_current(S_csr, raw_field(mp, "S"))
               and (not vel or (U_csr is not None and csr_is_current(U_csr, raw_field(mp, "U")))))

    def dense(t):
        t = f(t)
        return t.reshape(Ng, Nc) if tuple(t.shape) != (Ng, Nc) else t
    common = dict(
        kind=kind, noisemodel=mp.noisemodel, with_delta_nu=bool(mp.with_delta_nu),
        S=None if use_csr else dense(mp.S),
        count_factor=f(mp.count_factor).reshape(Nc), Db=f(mp.Db).reshape(int(mp.Nb), Nc),
        mu_nu=f(mp.μνg).reshape(Ng, -1), sd_nu=f(mp.σνg).reshape(Ng, -1),
        phixy_prior=f(mp.φxy_prior).reshape(Nc, 2), mu_dnu=float(mp.μΔν),
        gamma_alpha=float(mp.gamma_alpha), gamma_beta=float(mp.gamma_beta),
        condition_on={k: f(v) for k, v in condition_on.items()},
        S_csr=S_csr if use_csr else None)
    if kind == "phase":
        sd = f(mp.σΔν)
        return ModelSpec(guide="meanfield", H=int(mp.num_harmonics_S),
                         sd_dnu=(float(sd) if sd.numel() == 1 else sd.reshape(int(mp.Nb), Ng)),
                         sigma_ln_s=float(mp.σgc), **common)
    Nx = int(mp.Nx)
    return ModelSpec(
        guide=("lrmn" if mp.model_type == "lrmn" else "meanfield"), H=int(mp.num_harmonics),
        U=None if use_csr else dense(mp.U),
        D=f(mp.D).reshape(Nx, Nc), Hw=(int(mp.Nhω) - 1) // 2,
        mu_gamma=f(mp.μγ).reshape(Ng), sd_gamma=f(mp.σγ).reshape(Ng),
        mu_beta=f(mp.μβ).reshape(Ng), sd_beta=f(mp.σβ).reshape(Ng),
        mu_nuw=f(mp.μνω).reshape(Nx, -1), sd_nuw=f(mp.σνω).reshape(Nx, -1),
        sd_dnu=0.01, sigma_ln_s=float(mp.σsgc), sigma_ln_u=float(mp.σugc),
        rho_mean=float(mp.rho_mean), rho_std=float(mp.rho_std), rho_scale=float(mp.rho_scale),
        rho_rank=int(mp.rho_rank), U_csr=U_csr if use_csr else None, **common)
