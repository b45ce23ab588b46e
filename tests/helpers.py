"""Shared test helpers: fixtures -> oracle Problem / product ModelSpec."""
import glob
import os

import numpy as np
import torch

from oracle import velocycle_oracle as orc
from velocycle_amd.spec import ModelSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STEP_CASES = sorted(os.path.basename(p)[len("ref_step_"):-4] for p in glob.glob(os.path.join(GOLDEN, "ref_step_*.npz")))
FIT_CASES = sorted(os.path.basename(p)[len("ref_fit_"):-4] for p in glob.glob(os.path.join(GOLDEN, "ref_fit_*.npz")))

_TENSOR_FIELDS = ["S", "U", "count_factor", "Db", "D", "mu_nu", "sd_nu", "phixy_prior", "mu_gamma", "sd_gamma",
                  "mu_beta", "sd_beta", "mu_nuw", "sd_nuw"]
_SCALAR_FIELDS = ["kind", "guide", "noisemodel", "with_delta_nu", "H", "Hw", "mu_dnu", "gamma_alpha", "gamma_beta",
                  "sigma_ln_s", "sigma_ln_u", "rho_mean", "rho_std", "rho_scale", "rho_rank"]


def load_fixture(path):
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


def _fields(z, dtype):
    kw = {}
    for f in _TENSOR_FIELDS:
        if "in_" + f in z:
            kw[f] = torch.tensor(z["in_" + f]).to(dtype)
    for f in _SCALAR_FIELDS:
        if "in_" + f in z:
            v = z["in_" + f].item()
            kw[f] = v
    if "in_sd_dnu" in z:
        v = z["in_sd_dnu"]
        kw["sd_dnu"] = float(v) if v.ndim == 0 else torch.tensor(v).to(dtype)
    kw["condition_on"] = {k[len("cond_"):]: torch.tensor(v).to(dtype) for k, v in z.items() if k.startswith("cond_")}
    kw["with_delta_nu"] = bool(kw["with_delta_nu"])
    return kw


def problem_from_fixture(z, dtype=torch.float64) -> orc.Problem:
    return orc.Problem(**_fields(z, dtype))


def spec_from_fixture(z) -> ModelSpec:
    return ModelSpec(**_fields(z, torch.float32))


def spec_from_problem(p: orc.Problem) -> ModelSpec:
    kw = {}
    for k, v in p.__dict__.items():
        if isinstance(v, torch.Tensor):
            kw[k] = v.float()
        elif k == "condition_on":
            kw[k] = {a: b.float() for a, b in v.items()}
        else:
            kw[k] = v
    return ModelSpec(**kw)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    return float(np.abs(a - b).max() / scale)
