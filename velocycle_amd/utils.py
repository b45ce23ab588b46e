"""Host-side helpers with the reference's names (velocycle/utils.py:400-506, 586-610).  The engine
computes the basis on the GPU (K_pre); these torch versions serve result post-processing (ElogS /
ElogU summaries of fit()) and user code."""
from __future__ import annotations

import numpy as np
import torch


def torch_fourier_basis(ϕ, num_harmonics, der=0, device=torch.device("cpu")):
    """(Nc,) -> (Nc, 2H+1): [1, sin ϕ, cos ϕ, sin 2ϕ, ...] or its ϕ-derivative [0, cos ϕ, -sin ϕ, 2 cos 2ϕ, ...]."""
    if der not in (0, 1):
        raise ValueError(f"Value {der=} is not allowed, use 0 or 1 instead")
    ϕ = torch.as_tensor(ϕ).to(device)
    k = torch.arange(1, num_harmonics + 1, device=device, dtype=ϕ.dtype)
    arg = ϕ.unsqueeze(-1) * k                                           # (Nc, H)
    if der == 0:
        pair = torch.stack([torch.sin(arg), torch.cos(arg)], -1)
        head = torch.ones_like(ϕ)
    else:
        pair = torch.stack([k * torch.cos(arg), -k * torch.sin(arg)], -1)
        head = torch.zeros_like(ϕ)
    return torch.cat([head.unsqueeze(-1), pair.reshape(*ϕ.shape, 2 * num_harmonics)], -1).float()


def torch_basis(x, der=0, kind="fourier", device=torch.device("cpu"), **kwargs):
    if kind != "fourier":
        raise ValueError(f"{kind=} is not a valid entry use `fourier`")
    if "num_harmonics" not in kwargs:
        raise ValueError("num_harmonics needs to be provided if kind=`fourier`")
    return torch_fourier_basis(x, num_harmonics=kwargs["num_harmonics"], der=der, device=device).to(device)


def unpack_direction(loc, concentration=1.0):
    return torch.stack([torch.cos(loc), torch.sin(loc)], dim=-1) * concentration


def pack_direction(xy_pair):
    return torch.atan2(xy_pair[..., 1], xy_pair[..., 0])


def circular_corrcoef(x1, x2):
    assert len(x1) == len(x2), "Input arrays must have the same length"
    return float(np.abs(np.mean(np.exp(1j * np.asarray(x1)) * np.conj(np.exp(1j * np.asarray(x2))))))
