"""`preprocess_for_phase_estimation` / `preprocess_for_velocity_estimation`: build the
`MetaparContainer` record the fit drivers consume, with the reference's signatures and field names
(reference velocycle/preprocessing.py:20-323).  AnnData is duck-typed (`.layers`, `.obs`, `.var.index`,
`adata[:, genes]`): real `anndata.AnnData`, or `velocycle_amd.anndata_lite.AnnDataLite`.
"""
from __future__ import annotations

import unicodedata
from collections import defaultdict, namedtuple

import numpy as np
import torch

from .containers import AngularSpeed, Cycle, Phases, reorder  # noqa: F401


class _Program:
    """Placeholder for the Pyro program objects the reference stores in `model_fn` / `guide_fn`."""

    def __init__(self, name):
        self.__name__ = name

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self.__name__} is evaluated by the HIP engine (velocycle_amd), not as a Pyro program")

    def __repr__(self):
        return f"<velocycle_amd program {self.__name__}>"


phase_latent_variable_model = _Program("phase_latent_variable_model")
phase_latent_variable_guide = _Program("phase_latent_variable_guide")
velocity_latent_variable_model = _Program("velocity_latent_variable_model")
velocity_latent_variable_guide = _Program("velocity_latent_variable_guide")
velocity_latent_variable_model_LRMN = _Program("velocity_latent_variable_model_LRMN")
velocity_latent_variable_guide_LRMN = _Program("velocity_latent_variable_guide_LRMN")


def _dense(x):
    """scipy sparse / np.matrix / ndarray -> ndarray."""
    if hasattr(x, "toarray"):
        x = x.toarray()
    return np.asarray(x)


def _csr_counts(layer, dense=None):
    """scipy sparse layer -> canonical float32 CSR (cells x genes) with the reference's int64 truncation applied, or None
    for dense layers.  Carried in the container next to the dense S / U (which the reference's contract requires) so
    that the engine can ingest the sparse form directly (vc_set_counts_csr) without uploading the dense matrices.
    `dense` = the container's dense tensor of the same layer: the CSR copy is tagged with it (`csr_is_current`), so that
    a container whose S / U were replaced or edited afterwards never runs on stale CSR counts."""
    if not (hasattr(layer, "tocsr") and hasattr(layer, "toarray")):
        return None
    m = layer.tocsr().astype(np.int64).astype(np.float32)
    m.sum_duplicates()
    m.eliminate_zeros()
    if dense is not None:
        tag_csr(m, dense)
    return m


def tag_csr(csr, dense):
    """Record which dense tensor (object identity + in-place version counter) `csr` was derived from."""
    import weakref
    csr._vc_dense_ref = weakref.ref(dense)
    csr._vc_dense_version = dense._version
    return csr


def csr_is_current(csr, dense) -> bool:
    """True iff `csr` provably describes the same data as `dense`: `dense` is the very tensor object the CSR copy was
    built next to in `preprocess_for_*` and it has not been written in place since.  `mp._replace(S=...)`, in-place
    normalisation / subsampling / permutation of S all make this False, and the engine then reads the dense tensor."""
    ref = getattr(csr, "_vc_dense_ref", None)
    if csr is None or ref is None or dense is None:
        return False
    return ref() is dense and getattr(csr, "_vc_dense_version", -1) == dense._version


def _container(fields: dict):
    names = [unicodedata.normalize("NFKC", k) for k in fields]       # Python normalises identifiers
    return namedtuple("MetaparContainer", names)(*fields.values())


def filter_shared_genes(cycle, data, filter_type="intersection"):
    cycle_genes, data_genes = set(cycle.genes), set(data.var.index)
    if filter_type == "intersection":
        keep = np.array(sorted(cycle_genes & data_genes))
        return Cycle.from_array(means_array=cycle.means[keep], stds_array=cycle.stds[keep]), data[:, keep].copy()
    if filter_type == "union":
        if len(cycle_genes - data_genes) > 0:
            raise Exception("Gene features detected in Cycle object cannot be found in AnnData object")
        keep = np.array(sorted(cycle_genes | data_genes))
        new_cycle = Cycle.from_array(means_array=cycle.means, stds_array=cycle.stds)
        extra = np.array(sorted(data_genes - cycle_genes))
        if len(extra):
            import pandas as pd
            new_cycle.means = pd.concat([new_cycle.means, pd.DataFrame(0.0, index=new_cycle.means.index, columns=extra)], axis=1)
            new_cycle.stds = pd.concat([new_cycle.stds, pd.DataFrame(10.0, index=new_cycle.stds.index, columns=extra)], axis=1)
        return reorder(new_cycle, keep), data[:, keep].copy()
    raise Exception("Error: invalid argument for filter_type")


def make_design_matrix(anndata, ids="batch"):
    """(Nc, n_unique) int64 one-hot matrix, columns in order of first appearance (preprocessing.py:65-93)."""
    if ids not in anndata.obs.columns:
        raise ValueError(f"{ids=} is not a valid entry anndata.obs")
    seen = defaultdict(lambda: len(seen))
    codes = torch.tensor(np.array([seen[v] for v in np.array(anndata.obs[ids])]))
    return torch.stack([(codes == v).to(torch.int64) for v in range(len(seen))], dim=1)


def normalize_total(anndata):
    S, U = _dense(anndata.layers["spliced"]), _dense(anndata.layers["unspliced"])
    anndata.obs["n_scounts"], anndata.obs["n_ucounts"] = S.sum(1), U.sum(1)
    anndata.layers["S_sz"] = (np.mean(S.sum(1)) / S.sum(1) * S.T).T
    anndata.layers["U_sz"] = (np.mean(U.sum(1)) / U.sum(1) * U.T).T


def _counts_and_log(layer, truncate):
    """(float32 counts as a (Nc, Ng) tensor, float64 log(counts + 1 + 1e-16) as an ndarray) of one AnnData layer, with the
    values the reference gets from `torch.tensor(layer.astype(int64 | float))`, `.float()` and
    `np.log(S.numpy() + 1 + 1e-16)` (preprocessing.py:141-154, 243-268) but without its float64 / int64 copies of the
    matrix: 1.6 GB written per 50 000 x 2 000 layer instead of 5.6 GB.  truncate: the reference's int64 cast."""
    d = _dense(layer)
    if truncate and d.dtype.kind == "f":
        v = np.trunc(d)                                   # == astype(int64) for every count a float can hold exactly
        big = np.abs(v) >= 2.0 ** 62
        if big.any() or not np.isfinite(v).all():         # out of int64 range: keep numpy's own cast semantics
            v = d.astype(np.int64)
    else:
        v = d
    s32 = np.array(v, dtype=np.float32, order="C", copy=True)
    l64 = np.array(v, dtype=np.float64, order="C", copy=True)
    l64 += 1
    l64 += 1e-16
    np.log(l64, out=l64)
    return torch.from_numpy(s32), l64


def _t(x, device=None):
    return torch.as_tensor(x).float() if device is None else torch.as_tensor(x).float().to(device)


def preprocess_for_phase_estimation(anndata, cycle_obj, phase_obj, design_mtx, n_harmonics: int = 2,
                                    gene_selection_model: str = "all", normalize: bool = False,
                                    behavior: str = "intersection", noisemodel="NegativeBinomial",
                                    with_delta_nu: bool = True, condition_on={},
                                    μΔν=torch.tensor(0).float(), σΔν=torch.tensor(0.5).float(),
                                    gamma_alpha=torch.tensor(1.0).float(), gamma_beta=torch.tensor(2.0).float(),
                                    beta0=0.10, beta1=0.90, device=torch.device("cpu")):
    if gene_selection_model != "all":
        raise ValueError(f"{gene_selection_model=} is not a valid model")
    if normalize:
        if ("S_sz" not in anndata.layers) or ("U_sz" not in anndata.layers):
            normalize_total(anndata)
        (S, logS), (U, logU) = _counts_and_log(anndata.layers["S_sz"], False), _counts_and_log(anndata.layers["U_sz"], False)
    else:
        # preprocessing.py:141-147: layers with an `.A` attribute (scipy sparse, np.matrix) are cast to int64; a dense
        # ndarray has none, lands in the reference's `except` branch and stays float (non-integer values survive,
        # e.g. pre-normalised data for the Lognormal model)
        trunc = lambda layer: hasattr(layer, "A") or hasattr(layer, "toarray")
        S, logS = _counts_and_log(anndata.layers["spliced"], trunc(anndata.layers["spliced"]))
        U, logU = _counts_and_log(anndata.layers["unspliced"], trunc(anndata.layers["unspliced"]))
    s_umi = torch.tensor(np.asarray(_dense(anndata.layers["spliced"]).sum(1)).reshape(-1).astype(np.int64)).float()
    count_factor = torch.log(s_umi / torch.mean(s_umi))
    anndata.layers["logS"], anndata.layers["logU"] = logS, logU
    design_mtx = torch.as_tensor(design_mtx)
    S_t, U_t = S.T.to(device), U.T.to(device)                    # float32 views with strides (1, Ng), as `S.T.float()`
    fields = dict(
        Ng=len(cycle_obj), Nc=len(phase_obj), Nb=design_mtx.shape[-1],
        Db=design_mtx.T[:, None, :].float().to(device),
        cycle_prior=cycle_obj, phase_prior=phase_obj,
        μνg=cycle_obj.means_tensor.T[:, None, :].to(device), σνg=cycle_obj.stds_tensor.T[:, None, :].to(device),
        ϕxy_prior=phase_obj.phi_xy_tensor.T.to(device),
        gene_selection_model=gene_selection_model,
        model_fn=phase_latent_variable_model, guide_fn=phase_latent_variable_guide,
        num_harmonics_S=n_harmonics, basis_kind="fourier", noisemodel=noisemodel,
        gamma_alpha=_t(gamma_alpha, device), gamma_beta=_t(gamma_beta, device), device=device,
        kwargsζ=dict(num_harmonics=n_harmonics), σgc=torch.tensor(0.5).to(device),
        with_delta_nu=with_delta_nu, μΔν=_t(μΔν, device), σΔν=_t(σΔν, device),
        count_factor=count_factor[None, None, :].to(device),
        S=S_t, U=U_t,
        condition=np.array(list(condition_on.keys())),
        logS=torch.from_numpy(logS.astype(np.float32)).T.to(device),
        logU=torch.from_numpy(logU.astype(np.float32)).T.to(device),
        beta0=torch.tensor(beta0).to(device), beta1=torch.tensor(beta1).to(device),
        S_csr=None if normalize else _csr_counts(anndata.layers["spliced"], S_t),
        U_csr=None if normalize else _csr_counts(anndata.layers["unspliced"], U_t))
    return _container(fields)


def preprocess_for_velocity_estimation(anndata, cycle_obj, phase_obj, speed_obj, condition_design_mtx,
                                       batch_design_mtx, device=torch.device("cpu"),
                                       gene_selection_model: str = "all", null_cycle_obj=None,
                                       n_harmonics: int = 2, norm_size: int = 1000, with_delta_nu: bool = True,
                                       count_factor=0, count_factorU=0, ω_n_harmonics: int = 1,
                                       normalize: bool = False, behavior: str = "intersection",
                                       noisemodel="NegativeBinomial", condition_on={},
                                       μγ=torch.tensor(0.0).float(), σγ=torch.tensor(0.5).float(),
                                       μβ=torch.tensor(2.0).float(), σβ=torch.tensor(3.0).float(),
                                       μΔν=torch.tensor(0).float(), σΔν=torch.tensor(0.1).float(),
                                       gamma_alpha=torch.tensor(1.0).float(), gamma_beta=torch.tensor(2.0).float(),
                                       model_type: str = "lrmn", rho_mean=torch.tensor(4.0),
                                       rho_std=torch.tensor(1.0), rho_scale=torch.tensor(1.0),
                                       rho_rank=torch.tensor(5)):
    cycle_obj, anndata = filter_shared_genes(cycle_obj, anndata, filter_type=behavior)
    lay = ("S_sz", "U_sz") if normalize else ("spliced", "unspliced")
    S, logS = _counts_and_log(anndata.layers[lay[0]], True)
    U, logU = _counts_and_log(anndata.layers[lay[1]], True)
    if model_type == "lrmn":
        model_fn, guide_fn = velocity_latent_variable_model_LRMN, velocity_latent_variable_guide_LRMN
    elif gene_selection_model == "all":
        model_fn, guide_fn = velocity_latent_variable_model, velocity_latent_variable_guide
    else:
        raise ValueError(f"{gene_selection_model=} is not a valid model")
    anndata.layers["logS"], anndata.layers["logU"] = logS, logU
    ng = len(cycle_obj)
    cdm, bdm = torch.as_tensor(condition_design_mtx), torch.as_tensor(batch_design_mtx)
    rep = lambda v: torch.as_tensor(v).detach().clone().float().repeat([ng, 1]).to(device)
    S_t, U_t = S.T.to(device), U.T.to(device)
    fields = dict(
        Ng=ng, Nc=len(phase_obj), Nhω=(ω_n_harmonics * 2) + 1, Nb=bdm.shape[-1], Nx=cdm.shape[-1],
        D=cdm.T[:, None, None, :].clone().detach().to(device),
        Db=bdm.T[:, None, None, None, :].clone().detach().to(device),
        ν=cycle_obj.means_tensor.T.unsqueeze(-2).to(device),
        cycle_prior=cycle_obj, phase_prior=phase_obj, speed_prior=speed_obj,
        gene_selection_model=gene_selection_model, model_fn=model_fn, guide_fn=guide_fn,
        with_delta_nu=with_delta_nu, μΔν=_t(μΔν, device), σΔν=_t(σΔν, device),
        μγ=rep(μγ), σγ=rep(σγ), μβ=rep(μβ), σβ=rep(σβ),
        μνω=speed_obj.means_tensor.T.unsqueeze(-1).unsqueeze(-1).to(device),
        σνω=speed_obj.stds_tensor.T.unsqueeze(-1).unsqueeze(-1).to(device),
        μνg=cycle_obj.means_tensor.T[:, None, :].to(device), σνg=cycle_obj.stds_tensor.T[:, None, :].to(device),
        ϕxy_prior=phase_obj.phi_xy_tensor.T.to(device),
        basis_kind="fourier", num_harmonics=n_harmonics, noisemodel=noisemodel,
        gamma_alpha=_t(gamma_alpha, device), gamma_beta=_t(gamma_beta, device),
        count_factor=torch.as_tensor(count_factor).clone().detach().to(device),
        kwargsζ=dict(num_harmonics=n_harmonics), kwargsζ_dϕ=dict(num_harmonics=n_harmonics),
        kwargsζω=dict(num_harmonics=ω_n_harmonics),
        σₛgc=torch.tensor(0.1, device=device), σᵤgc=torch.tensor(0.1, device=device),
        S=S_t, U=U_t,
        logS=torch.from_numpy(logS.astype(np.float32)).T.to(device),
        logU=torch.from_numpy(logU.astype(np.float32)).T.to(device),
        condition=np.array(list(condition_on.keys())), device=device, model_type=model_type,
        rho_mean=torch.as_tensor(rho_mean).to(device), rho_std=torch.as_tensor(rho_std).to(device),
        rho_scale=torch.as_tensor(rho_scale).to(device), rho_rank=torch.as_tensor(rho_rank).to(device),
        S_csr=None if normalize else _csr_counts(anndata.layers[lay[0]], S_t),
        U_csr=None if normalize else _csr_counts(anndata.layers[lay[1]], U_t))
    return _container(fields)
