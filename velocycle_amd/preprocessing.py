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
    src = layer.tocsr()
    # the values in one or two passes over the data array alone (scipy's astype copies the index arrays with every call), the
    # index arrays copied once: the container's CSR never aliases the caller's matrix
    data = src.data if src.data.dtype.kind in "iub" else src.data.astype(np.int64)
    m = type(src)((data.astype(np.float32), src.indices.copy(), src.indptr.copy()), shape=src.shape)
    m.sum_duplicates()
    m.eliminate_zeros()
    if dense is not None:
        tag_csr(m, dense)
    return m


class _Lazy:
    """A dense container field of a sparse layer that is built when somebody reads it (`mp.S`, `mp.logU`, ...): the
    reference's contract wants the dense float32 views in the container, the HIP engine ingests the CSR form and never
    looks at them -- at 50 000 x 2 000 building them is half the wall time of a whole fit().  Reading the attribute gives
    the same tensor the eager code built, bit for bit; index access (`mp[i]`) and iteration see this placeholder."""
    __slots__ = ("fn", "value", "done", "version0", "__weakref__")

    def __init__(self, fn):
        self.fn, self.value, self.done, self.version0 = fn, None, False, None

    def __call__(self):
        if not self.done:
            self.value = self.fn()
            self.fn = None
            self.version0 = getattr(self.value, "_version", None)
            self.done = True
        return self.value


def raw_field(mp, name):
    """The stored item of a container field: the tensor, or its `_Lazy` placeholder (without building it)."""
    try:
        return tuple.__getitem__(mp, mp._fields.index(unicodedata.normalize("NFKC", name)))
    except (ValueError, AttributeError, TypeError):
        return getattr(mp, name, None)


def tag_csr(csr, dense):
    """Record which dense field (tensor or `_Lazy` placeholder: object identity + in-place version counter) `csr`
    was derived next to."""
    import weakref
    csr._vc_dense_ref = weakref.ref(dense)
    csr._vc_dense_version = getattr(dense, "_version", None)
    return csr


def csr_is_current(csr, dense) -> bool:
    """True iff `csr` provably describes the same data as the container's dense field `dense` (the tensor, or the raw
    `_Lazy` placeholder from `raw_field`): it is the very object the CSR copy was built next to in `preprocess_for_*` and
    it has not been written in place since.  `mp._replace(S=...)`, in-place normalisation / subsampling / permutation of
    S all make this False, and the engine then reads the dense tensor."""
    ref = getattr(csr, "_vc_dense_ref", None)
    if csr is None or ref is None or dense is None:
        return False
    if isinstance(dense, _Lazy):
        return ref() is dense and (not dense.done or getattr(dense.value, "_version", None) == dense.version0)
    tagged = ref()
    if isinstance(tagged, _Lazy):         # the caller handed over the materialised tensor of the tagged placeholder
        return tagged.done and tagged.value is dense and dense._version == tagged.version0
    return tagged is dense and getattr(csr, "_vc_dense_version", -1) == dense._version


_CONTAINER_CLASSES = {}


def _container(fields: dict):
    names = tuple(unicodedata.normalize("NFKC", k) for k in fields)       # Python normalises identifiers
    lazy = tuple(i for i, v in enumerate(fields.values()) if isinstance(v, _Lazy))
    cls = _CONTAINER_CLASSES.get((names, lazy))
    if cls is None:
        base = namedtuple("MetaparContainer", names)
        ns = {"__slots__": ()}
        for i in lazy:             # attribute access builds the field; `_replace`, `_make`, pickling carry the placeholder
            def get(self, _i=i):
                v = tuple.__getitem__(self, _i)
                return v() if isinstance(v, _Lazy) else v
            ns[names[i]] = property(get, doc=f"field {names[i]} (dense view of a sparse layer, built on first read)")
        cls = type("MetaparContainer", (base,), ns)
        _CONTAINER_CLASSES[(names, lazy)] = cls
    return cls(*fields.values())


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


def _is_sparse_layer(layer):
    return hasattr(layer, "tocsr") and hasattr(layer, "toarray")


class _LayerFields:
    """The four dense container fields derived from one AnnData layer -- counts (Ng, Nc) float32 view, log counts as
    float32 (Ng, Nc), the float64 log layer the reference leaves in `anndata.layers` -- built together on first use
    (sparse layers) or at once (dense layers, as before)."""

    def __init__(self, layer, truncate, device, lazy):
        self._args, self._res = (layer, truncate, device), None
        if not lazy:
            self._build()

    def _build(self):
        if self._res is None:
            layer, truncate, device = self._args
            S, logS = _counts_and_log(layer, truncate)
            self._res = (S.T.to(device), logS, torch.from_numpy(logS.astype(np.float32)).T.to(device))
            self._args = None
        return self._res

    def counts(self):
        return self._build()[0]

    def log64(self):
        return self._build()[1]

    def log32(self):
        return self._build()[2]

    def field(self, which, lazy):
        fn = {"counts": self.counts, "log32": self.log32, "log64": self.log64}[which]
        return _Lazy(fn) if lazy else fn()


def _set_layer(anndata, key, fields: _LayerFields, lazy):
    """anndata.layers[key] = the float64 log layer (preprocessing.py:153-154); on an AnnDataLite with sparse input it is
    installed as a lazy layer (built when read), a real AnnData gets the array."""
    if lazy and hasattr(anndata.layers, "set_lazy"):
        anndata.layers.set_lazy(key, fields.log64)
    else:
        anndata.layers[key] = fields.log64()


def _row_sums_int(layer):
    """Per-cell UMI totals of a layer with the reference's int64 cast, without forming the dense matrix of a sparse one."""
    if _is_sparse_layer(layer):      # sum in the layer's own dtype, then the cast (preprocessing.py:149)
        return np.asarray(layer.sum(1)).reshape(-1).astype(np.int64)
    return np.asarray(_dense(layer).sum(1)).reshape(-1).astype(np.int64)


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
        lazy = False
        fS, fU = _LayerFields(anndata.layers["S_sz"], False, device, False), _LayerFields(anndata.layers["U_sz"], False, device, False)
    else:
        # preprocessing.py:141-147: layers with an `.A` attribute (scipy sparse, np.matrix) are cast to int64; a dense
        # ndarray has none, lands in the reference's `except` branch and stays float (non-integer values survive,
        # e.g. pre-normalised data for the Lognormal model)
        trunc = lambda layer: hasattr(layer, "A") or hasattr(layer, "toarray")
        lS, lU = anndata.layers["spliced"], anndata.layers["unspliced"]
        lazy = _is_sparse_layer(lS) and _is_sparse_layer(lU)       # sparse layers: the engine ingests the CSR form
        fS, fU = _LayerFields(lS, trunc(lS), device, lazy), _LayerFields(lU, trunc(lU), device, lazy)
    s_umi = torch.tensor(_row_sums_int(anndata.layers["spliced"])).float()
    count_factor = torch.log(s_umi / torch.mean(s_umi))
    _set_layer(anndata, "logS", fS, lazy)
    _set_layer(anndata, "logU", fU, lazy)
    design_mtx = torch.as_tensor(design_mtx)
    S_t, U_t = fS.field("counts", lazy), fU.field("counts", lazy)   # float32 views with strides (1, Ng), as `S.T.float()`
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
        logS=fS.field("log32", lazy), logU=fU.field("log32", lazy),
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
    lS, lU = anndata.layers[lay[0]], anndata.layers[lay[1]]
    lazy = (not normalize) and _is_sparse_layer(lS) and _is_sparse_layer(lU)
    fS, fU = _LayerFields(lS, True, device, lazy), _LayerFields(lU, True, device, lazy)
    if model_type == "lrmn":
        model_fn, guide_fn = velocity_latent_variable_model_LRMN, velocity_latent_variable_guide_LRMN
    elif gene_selection_model == "all":
        model_fn, guide_fn = velocity_latent_variable_model, velocity_latent_variable_guide
    else:
        raise ValueError(f"{gene_selection_model=} is not a valid model")
    _set_layer(anndata, "logS", fS, lazy)
    _set_layer(anndata, "logU", fU, lazy)
    ng = len(cycle_obj)
    cdm, bdm = torch.as_tensor(condition_design_mtx), torch.as_tensor(batch_design_mtx)
    rep = lambda v: torch.as_tensor(v).detach().clone().float().repeat([ng, 1]).to(device)
    S_t, U_t = fS.field("counts", lazy), fU.field("counts", lazy)
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
        logS=fS.field("log32", lazy), logU=fU.field("log32", lazy),
        condition=np.array(list(condition_on.keys())), device=device, model_type=model_type,
        rho_mean=torch.as_tensor(rho_mean).to(device), rho_std=torch.as_tensor(rho_std).to(device),
        rho_scale=torch.as_tensor(rho_scale).to(device), rho_rank=torch.as_tensor(rho_rank).to(device),
        S_csr=None if normalize else _csr_counts(anndata.layers[lay[0]], S_t),
        U_csr=None if normalize else _csr_counts(anndata.layers[lay[1]], U_t))
    return _container(fields)
