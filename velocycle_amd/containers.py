"""Result / prior containers with the reference's public surface (data holders only):
`Cycle` (reference velocycle/cycle.py:10-466), `AngularSpeed` (angularspeed.py:10-354),
`Phases` (phases.py:103-305).  They are what `fit()` returns and what `preprocess_for_*` consume.

Shared design here: Cycle and AngularSpeed are both "a (2H+1) x N table of means plus one of stds",
so both derive from `_HarmonicTable`.  Quirk kept on purpose: `from_array` labels the
rows nu0, nu1_cos, nu1_sin, ... although the arithmetic order of the basis is [1, sin, cos, ...].
On-disk format (`save`/`load`): one CSV, the means table stacked over the stds table.
"""
from __future__ import annotations

import copy as _copy

import numpy as np
import pandas as pd
import torch


def _row_labels(n_rows: int, swapped: bool):
    first, second = ("cos", "sin") if swapped else ("sin", "cos")
    return ["nu0"] + [f"nu{i // 2 + 1}_{second if i % 2 else first}" for i in range(n_rows - 1)]


def _as_frame(new, like: pd.DataFrame) -> pd.DataFrame:
    if isinstance(new, pd.DataFrame):
        return new
    if torch.is_tensor(new):
        new = new.detach().cpu().numpy()
    if isinstance(new, np.ndarray):
        return pd.DataFrame(new, index=like.index, columns=like.columns)
    raise Exception("Error: invalid type for new values")


class _HarmonicTable:
    def __init__(self):
        self.means: pd.DataFrame = None
        self.stds: pd.DataFrame = None

    def __len__(self):
        return self.shape[-1]

    def __getitem__(self, key):
        out = type(self)()
        out.means = self.means.__getitem__(key)
        out.stds = self.stds.__getitem__(key)
        return out

    def set_means(self, new_means):
        self.means = _as_frame(new_means, self.means)

    def set_stds(self, new_stds):
        self.stds = _as_frame(new_stds, self.stds)

    @property
    def harmonics(self):
        return (self.means.shape[0] - 1) // 2

    @property
    def shape(self):
        return self.means.shape

    @property
    def means_tensor(self):
        return torch.tensor(self.means.values.astype(np.float32))

    @property
    def stds_tensor(self):
        return torch.tensor(self.stds.values.astype(np.float32))

    def save(self, pathname):
        pd.concat([self.means, self.stds]).to_csv(pathname)

    @classmethod
    def load(cls, filepath):
        df = pd.read_csv(filepath, index_col=0)
        half = df.shape[0] // 2
        out = cls()
        out.means, out.stds = df.iloc[:half, :], df.iloc[half:, :]
        return out

    from_file = load

    def copy(self):
        return _copy.deepcopy(self)


class Cycle(_HarmonicTable):
    """Fourier coefficients of every gene: means/stds are (2H+1) x Ng DataFrames."""

    def __init__(self):
        super().__init__()
        self.log_gammas = None
        self.log_betas = None
        self.disp_pyro = None
        self.periodic = None

    def set_log_gammas(self, v):
        self.log_gammas = v

    def set_log_betas(self, v):
        self.log_betas = v

    def set_disp_pyro(self, v):
        self.disp_pyro = v

    @property
    def genes(self):
        return list(self.means.columns)

    @classmethod
    def from_array(cls, means_array, stds_array, gene_names=None):
        means_array, stds_array = np.asarray(means_array), np.asarray(stds_array)
        assert means_array.shape == stds_array.shape, "Shapes of the arrays must be equal"
        if gene_names is not None:
            assert len(gene_names) == means_array.shape[1]
        rows = _row_labels(means_array.shape[0], swapped=True)          # reference cycle.py:321-323
        out = cls()
        out.means = pd.DataFrame(means_array, index=rows, columns=gene_names)
        out.stds = pd.DataFrame(stds_array, index=rows, columns=gene_names)
        return out

    @classmethod
    def trivial_prior(cls, gene_names, harmonics=2, means=0.0, stds=3.0):
        if harmonics == 1:
            stds = np.array([.1, .2, .2])[:, None]
        if harmonics == 2:
            stds = np.array([.1, .2, .2, .1, .1])[:, None]
        nrow = 2 * harmonics + 1
        rows = _row_labels(nrow, swapped=True)
        out = cls()
        out.means = pd.DataFrame(np.broadcast_to(means, (nrow, len(gene_names))).copy(), index=rows, columns=gene_names)
        out.stds = pd.DataFrame(np.broadcast_to(stds, (nrow, len(gene_names))).copy(), index=rows, columns=gene_names)
        return out


def reorder(cycle: Cycle, gene_list):
    return Cycle.from_array(means_array=cycle.means[gene_list], stds_array=cycle.stds[gene_list])


class AngularSpeed(_HarmonicTable):
    """Fourier coefficients of the angular speed of every condition: (2Hw+1) x Nx DataFrames."""

    @property
    def conditions(self):
        return list(self.means.columns)

    @classmethod
    def from_array(cls, means_array, stds_array, condition_names=None, Nhω=0):
        means_array, stds_array = np.asarray(means_array), np.asarray(stds_array)
        assert means_array.shape == stds_array.shape, "Shapes of the arrays must be equal"
        rows = _row_labels(max(int(Nhω), 1), swapped=True)

        def table(a):
            df = pd.DataFrame([a]) if len(rows) == 1 else pd.DataFrame(np.asarray(a).squeeze())
            if len(df.index) == len(rows):
                df.index, df.columns = rows, condition_names
                return df
            df.index, df.columns = condition_names, rows
            return df.T
        out = cls()
        out.means, out.stds = table(means_array), table(stds_array)
        return out

    @classmethod
    def trivial_prior(cls, condition_names, harmonics=1, means=0.0, stds=3.0):
        nrow = 2 * harmonics + 1
        rows = _row_labels(nrow, swapped=True)
        mu = np.array([means] + [0.0] * (nrow - 1), dtype=np.float32)[:, None]
        sd = np.array([stds] + [0.05] * (nrow - 1), dtype=np.float32)[:, None]
        out = cls()
        out.means = pd.DataFrame(np.broadcast_to(mu, (nrow, len(condition_names))).copy(), index=rows, columns=condition_names)
        out.stds = pd.DataFrame(np.broadcast_to(sd, (nrow, len(condition_names))).copy(), index=rows, columns=condition_names)
        return out


class Phases:
    """Cell phases as 2 x Nc direction vectors (rows phi_x, phi_y)."""

    def __init__(self):
        self.phi_xy: pd.DataFrame = None
        self.omegas = None

    def __len__(self):
        return self.shape[-1]

    @property
    def shape(self):
        return self.phi_xy.shape

    def set_phixy(self, new_phixy):
        self.phi_xy = _as_frame(new_phixy, self.phi_xy)

    def set_omegas(self, new_omegas):
        self.omegas = new_omegas

    @property
    def phi_xy_tensor(self):
        return torch.tensor(self.phi_xy.values.astype(np.float32))

    @property
    def phis(self):
        xy = self.phi_xy_tensor.T
        phis = torch.atan2(xy[..., 1], xy[..., 0])
        phis[phis < 0] = phis[phis < 0] + 2 * np.pi
        return phis

    @property
    def directions(self):
        return np.arctan2(self.phi_xy.values[1, :], self.phi_xy.values[0, :]) % (2 * np.pi)

    @property
    def concentrations(self):
        return np.sqrt(np.sum(self.phi_xy.values ** 2, 0))

    @classmethod
    def from_array(cls, phi_xy_array, cell_names=None):
        phi_xy_array = np.asarray(phi_xy_array)
        assert phi_xy_array.shape[0] == 2, "Shape of the array is incorrect"
        if cell_names is not None:
            assert len(cell_names) == phi_xy_array.shape[1]
        out = cls()
        out.phi_xy = pd.DataFrame(phi_xy_array, index=["phi_x", "phi_y"], columns=cell_names)
        return out

    @classmethod
    def from_pca_heuristic(cls, anndata_object, genes_to_use=None, concentration=1.0, layer="S_sz", small_count=1.0e-1,
                           normalize_pcs=True, zero_at_min_density=False, random_state=0, plot=False, n_components=2):
        """Phase prior from the angle in the plane of the first two principal components of the log counts
        (reference phases.py:307-382; the step right before phase inference in the tutorials)."""
        from sklearn.decomposition import PCA
        if layer not in anndata_object.layers:
            raise ValueError(f"{layer=} is not a valid entry anndata.obs")
        sub = anndata_object if genes_to_use is None else \
            anndata_object[:, [g in genes_to_use for g in anndata_object.var.index]]
        mat = sub.layers[layer]
        mat = mat.toarray() if hasattr(mat, "toarray") else np.asarray(mat)
        X = np.log(mat + small_count)                                   # cells x genes
        pca = PCA(n_components, random_state=random_state)
        pcs = pca.fit_transform(X)
        if normalize_pcs:
            lo, hi, med = np.percentile(pcs, [0.5, 99.5, 50], 0)
            pcs = (pcs - med) / (hi - lo)
        angle = np.arctan2(pcs[:, 1], pcs[:, 0]) % (2 * np.pi)
        if zero_at_min_density:          # put phase 0 right after the widest gap of the sorted angles
            order = np.argsort(angle)
            start = order[np.diff(angle[order]).argmax() + 1]
            angle = (angle - angle[start]) % (2 * np.pi)
        out = cls.from_array(np.vstack([np.cos(angle), np.sin(angle)]) * concentration,
                             cell_names=anndata_object.obs.index)
        out.pcs, out.pca = pcs, pca
        return out

    def max_corr(self, counts, npoints=100):
        """Shift (out of `npoints` equally spaced ones) maximising the correlation of the phases with `counts`
        (reference phases.py:450-469).  Returns (shift, correlation, all correlations)."""
        shifts = np.arange(0, npoints) / npoints * 2 * np.pi
        phis = self.phis.numpy()
        corr = []
        for s in shifts:
            x = phis - s
            x[x < 0] += 2 * np.pi
            corr.append(np.corrcoef(x, counts)[0, 1])
        i = int(np.argmax(np.array(corr)))
        return shifts[i], corr[i], corr

    def shift_zero(self, gene=None, phase=None):
        if gene is not None:
            raise Exception("Error: must phase for desired shift")
        if phase is None:
            raise Exception("Error: must specify gene or phase for desired shift")
        ph = self.phis - phase
        self.set_phixy(torch.stack([torch.cos(ph), torch.sin(ph)], dim=-1).T)

    @classmethod
    def flat_prior(cls, anndata_object):
        n = anndata_object.shape[0]
        return cls.from_array(np.zeros((2, n)), cell_names=list(anndata_object.obs.index))

    def rotate(self, angle=None):
        if angle is None:
            raise Exception("Error: must specify angle for desired rotation")
        c, s = np.cos(angle), np.sin(angle)
        rot = np.array([[c, -s], [s, c]])
        self.phi_xy = pd.DataFrame(rot @ self.phi_xy.values, index=self.phi_xy.index, columns=self.phi_xy.columns)

    def invert_direction(self):
        self.phi_xy = pd.DataFrame(self.phi_xy.values * np.array([[1.0], [-1.0]]), index=self.phi_xy.index,
                                   columns=self.phi_xy.columns)

    def save(self, pathname):
        self.phi_xy.to_csv(pathname)

    @classmethod
    def load(cls, filepath):
        out = cls()
        out.phi_xy = pd.read_csv(filepath, index_col=0)
        return out

    from_file = load
