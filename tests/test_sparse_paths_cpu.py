"""Host-side sparse paths of the preprocess stage (reference: velocycle/preprocessing.py:120-122, 274-276 -- the layers go through
`.toarray().astype(int64)` before they become float32 tensors): the container's CSR copy and AnnDataLite's selection keep their
results when they take the cheap route (one pass over the values, a copy instead of a fancy index)."""
import numpy as np
import pytest
import scipy.sparse as sp

from velocycle_amd.anndata_lite import AnnDataLite
from velocycle_amd.preprocessing import _csr_counts


def _reference_dense(layer):
    return layer.toarray().astype(np.int64).astype(np.float32)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int32, np.uint16])
def test_csr_counts_is_the_truncated_layer(dtype):
    rng = np.random.default_rng(3)
    dense = rng.poisson(0.7, size=(40, 23)).astype(np.float64)
    if np.dtype(dtype).kind == "f":
        dense += rng.random(dense.shape) * (rng.random(dense.shape) < 0.3)        # fractional counts are truncated, 0.x -> an explicit zero
    layer = sp.csr_matrix(dense.astype(dtype))
    got = _csr_counts(layer)
    assert got.dtype == np.float32 and got.has_canonical_format
    assert np.array_equal(got.toarray(), _reference_dense(layer))
    assert got.nnz == np.count_nonzero(_reference_dense(layer))                   # no stored zeros
    assert not np.shares_memory(got.indices, layer.indices) and not np.shares_memory(got.indptr, layer.indptr)


def test_csr_counts_sums_duplicates_and_takes_other_formats():
    rows, cols, vals = [0, 0, 1, 2, 2], [1, 1, 0, 2, 2], [1.0, 2.0, 3.0, 0.6, 0.7]
    coo = sp.coo_matrix((vals, (rows, cols)), shape=(3, 3))
    got = _csr_counts(coo)
    want = coo.tocsr().astype(np.int64).astype(np.float32)
    want.sum_duplicates(); want.eliminate_zeros()
    assert np.array_equal(got.toarray(), want.toarray()) and got.nnz == want.nnz
    assert _csr_counts(np.ones((2, 2))) is None                                   # dense layers have no CSR copy


@pytest.mark.parametrize("rows", ["all", "some"])
@pytest.mark.parametrize("cols", ["all", "some"])
def test_selected_sparse_layers_equal_the_fancy_index(rows, cols):
    rng = np.random.default_rng(5)
    S = sp.csr_matrix(rng.poisson(0.5, size=(30, 12)).astype(np.float32))
    U = sp.csr_matrix(rng.poisson(0.2, size=(30, 12)).astype(np.float32))
    ad = AnnDataLite(S, U)
    r = np.arange(30) if rows == "all" else np.array([3, 1, 7, 29])
    c = np.arange(12) if cols == "all" else np.array([11, 0, 5])
    sub = ad[ad.obs.index[r], ad.var.index[c]] if hasattr(ad.obs, "index") else ad[r, c]
    for name, m in (("spliced", S), ("unspliced", U)):
        got = sub.layers[name]
        assert np.array_equal(got.toarray(), m.toarray()[np.ix_(r, c)])
        assert not np.shares_memory(got.data, m.data)                             # a selection is a copy
