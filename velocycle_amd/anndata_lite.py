"""A tiny stand-in for the slice of the AnnData interface that the reference's `preprocess_for_*`
touch (reference velocycle/preprocessing.py:20-63,124-154,241-268): `.layers[...]`, `.obs[...]`,
`.var.index`, `adata[:, genes].copy()`.  `anndata`/`h5py` are not installed in this image; real
`anndata.AnnData` objects are accepted everywhere this class is."""
from __future__ import annotations

from collections.abc import MutableMapping

import numpy as np
import pandas as pd


class AnnDataLite:
    def __init__(self, spliced, unspliced, gene_names=None, cell_names=None, obs=None):
        # dense arrays or scipy sparse matrices (what .h5ad files usually hold); sparse layers stay sparse
        spliced = spliced if _is_sparse(spliced) else np.asarray(spliced)
        unspliced = unspliced if _is_sparse(unspliced) else np.asarray(unspliced)
        assert spliced.shape == unspliced.shape and len(spliced.shape) == 2
        nc, ng = spliced.shape
        if gene_names is None:
            gene_names = ["G" + str(i).zfill(5) for i in range(ng)]
        if cell_names is None:
            cell_names = ["C" + str(i).zfill(6) for i in range(nc)]
        self.layers = _Layers({"spliced": spliced, "unspliced": unspliced})
        self.var = pd.DataFrame(index=pd.Index(list(gene_names)))
        self.obs = pd.DataFrame(index=pd.Index(list(cell_names))) if obs is None else obs
        self.X = spliced

    @property
    def shape(self):
        return self.layers["spliced"].shape

    @property
    def n_obs(self):
        return self.shape[0]

    @property
    def n_vars(self):
        return self.shape[1]

    def copy(self):
        if isinstance(self.layers, _SelectedLayers):      # a fresh adata[cells, genes]: its layers are copies already
            return self
        out = AnnDataLite.__new__(AnnDataLite)
        out.layers = _Layers()
        for k in self.layers.keys():
            v = self.layers._raw(k) if isinstance(self.layers, _Layers) else self.layers[k]
            if isinstance(v, _LazyLayer):                  # not built yet: the copy builds its own when somebody reads it
                out.layers.set_lazy(k, v.fn)
            else:
                out.layers[k] = v.copy() if _is_sparse(v) else np.array(v, copy=True)
        out.var = self.var.copy()
        out.obs = self.obs.copy()
        out.X = out.layers["spliced"]
        return out

    def __getitem__(self, key):
        if not (isinstance(key, tuple) and len(key) == 2):
            raise IndexError("AnnDataLite supports adata[cells, genes] only")
        rows, cols = key
        ridx = np.arange(self.n_obs)[rows] if isinstance(rows, slice) else \
            self.obs.index.get_indexer(list(rows)) if _is_names(rows) else np.asarray(rows)
        cidx = np.arange(self.n_vars)[cols] if isinstance(cols, slice) else \
            self.var.index.get_indexer(list(cols)) if _is_names(cols) else np.asarray(cols)
        if (np.asarray(cidx) < 0).any() or (np.asarray(ridx) < 0).any():
            raise KeyError("unknown gene / cell name")
        out = AnnDataLite.__new__(AnnDataLite)
        out.layers = _SelectedLayers(self.layers, ridx, cidx)
        out.var = self.var.iloc[cidx].copy()
        out.obs = self.obs.iloc[ridx].copy()
        return out

    @property
    def X(self):
        return self.layers["spliced"]

    @X.setter
    def X(self, v):
        pass


class _LazyLayer:
    def __init__(self, fn):
        self.fn = fn


class _Layers(MutableMapping):
    """`adata.layers`: a mapping whose entries may be installed lazily (`set_lazy(key, fn)`: built when first read) -- the
    float64 logS / logU layers that `preprocess_for_*` leave behind are 800 MB each at 50 000 x 2 000 and are read by plots
    only.  A MutableMapping (not a dict subclass): every way of reading an entry -- `[]`, get, items, values, pop, setdefault,
    dict(layers), {**layers}, copy() -- goes through `__getitem__`, so a placeholder never leaks out."""

    def __init__(self, init=None):
        self._d = {}
        if init is not None:
            for k in init.keys():
                self._d[k] = init._raw(k) if isinstance(init, _Layers) else init[k]

    def set_lazy(self, key, fn):
        self._d[key] = _LazyLayer(fn)

    def _raw(self, k):
        """The stored entry, a `_LazyLayer` placeholder included (copies of the mapping carry those over unbuilt)."""
        return self._d[k]

    def __getitem__(self, k):
        v = self._d[k]
        if isinstance(v, _LazyLayer):
            v = v.fn()
            self._d[k] = v
        return v

    def __setitem__(self, k, v):
        self._d[k] = v

    def __delitem__(self, k):
        del self._d[k]

    def __iter__(self):
        return iter(self._d)

    def __len__(self):
        return len(self._d)

    def __contains__(self, k):
        return k in self._d

    def __repr__(self):
        return "Layers(" + ", ".join(repr(k) + (" (lazy)" if isinstance(v, _LazyLayer) else "") for k, v in self._d.items()) + ")"

    def copy(self):
        """Shallow copy like dict.copy(); lazily installed entries stay lazy."""
        return _Layers(self)

    def __reduce__(self):
        return (_Layers, (dict(self),))          # pickling materialises: a closure does not travel


class _SelectedLayers(_Layers):
    """Layers of adata[cells, genes]: every layer is selected (= copied) the first time it is read, so that layers nobody
    asks for -- the float64 logS / logU a phase preprocess left behind, 800 MB each at 50 000 x 2 000 -- are never copied."""

    def __init__(self, parent, ridx, cidx):
        super().__init__()
        # (a snapshot of the parent's entries; lazily installed ones stay lazy until this selection reads them)
        self._parent = {k: (parent._raw(k) if isinstance(parent, _Layers) else parent[k]) for k in parent.keys()}
        self._ridx, self._cidx = np.asarray(ridx), np.asarray(cidx)

    def _take(self, v):
        if isinstance(v, _LazyLayer):
            v = v.fn()
        if _is_sparse(v):
            # (a selection that keeps every row / every column in order costs a copy, not a fancy index: at 50 000 x 2 000
            # scipy's column gather alone is 0.1 s per layer)
            m = v.tocsr()
            rows_all = len(self._ridx) == m.shape[0] and np.array_equal(self._ridx, np.arange(m.shape[0]))
            cols_all = len(self._cidx) == m.shape[1] and np.array_equal(self._cidx, np.arange(m.shape[1]))
            if rows_all and cols_all:
                return m.copy()
            if not rows_all:
                m = m[self._ridx]
            return m if cols_all else m[:, self._cidx]
        v = np.asarray(v)
        if len(self._ridx) == v.shape[0] and np.array_equal(self._ridx, np.arange(v.shape[0])):
            return np.take(v, self._cidx, axis=1)        # all cells: one gather along the genes
        return v[np.ix_(self._ridx, self._cidx)]

    def _raw(self, k):
        if k in self._d:
            return self._d[k]
        parent_entry = self._parent[k]                    # KeyError if the parent has no such layer either
        return _LazyLayer(lambda: self._take(parent_entry))

    def __getitem__(self, k):
        if k not in self._d:
            self._d[k] = self._take(self._parent[k])      # KeyError if the parent has no such layer either
        return super().__getitem__(k)

    def __delitem__(self, k):
        found = False
        if k in self._d:
            del self._d[k]
            found = True
        if k in self._parent:
            del self._parent[k]
            found = True
        if not found:
            raise KeyError(k)

    def __contains__(self, k):
        return k in self._d or k in self._parent

    def __iter__(self):
        return iter(dict.fromkeys(list(self._parent) + list(self._d)))

    def __len__(self):
        return len(dict.fromkeys(list(self._parent) + list(self._d)))

    def copy(self):
        return _Layers(self)

    def __reduce__(self):
        return (_Layers, (dict(self),))


def _is_sparse(x):
    return hasattr(x, "toarray") and hasattr(x, "tocsr")


def _is_names(k):
    k = np.asarray(k)
    return k.dtype.kind in "OUS"
