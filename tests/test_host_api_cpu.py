"""CPU: host-side mirror of the reference's Python surface -- containers, design matrix and the
MetaparContainer built by preprocess_for_* -- against tensors produced by the reference's own
preprocessing (tests/golden/ref_preprocess.npz), plus the eps stream order and ClippedAdam restatement."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import velocycle_oracle as orc
from tests import helpers as H
from velocycle_amd import containers as C
from velocycle_amd import preprocessing as P
from velocycle_amd.anndata_lite import AnnDataLite


def _inputs():
    z = H.load_fixture(f"{H.GOLDEN}/ref_preprocess.npz")
    ad = AnnDataLite(z["S"], z["U"])
    ad.obs["batch"] = list(z["batch"])
    genes = list(ad.var.index)
    cyc = C.Cycle.trivial_prior(genes, harmonics=1)
    cyc.set_means(z["cyc_means"])
    cyc.set_stds(z["cyc_stds"])
    ph = C.Phases.from_array(z["phi_xy"], cell_names=list(ad.obs.index))
    return z, ad, cyc, ph


def test_design_matrix_and_phase_container_match_reference():
    z, ad, cyc, ph = _inputs()
    Db = P.make_design_matrix(ad, ids="batch")
    assert Db.dtype == torch.int64 and np.array_equal(Db.numpy(), z["design"])
    with pytest.raises(ValueError):
        P.make_design_matrix(ad, ids="nope")
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1)
    for k in ("Db", "μνg", "σνg", "φxy_prior", "count_factor", "S", "U", "logS", "σΔν", "μΔν"):
        got = getattr(mp, k).numpy()
        assert got.shape == z["phase_" + k].shape, k
        assert np.allclose(got, z["phase_" + k], rtol=1e-6, atol=1e-6), k
    assert mp.S.stride() == (1, mp.Ng)                       # the reference's transposed view (cell-major)
    assert mp.model_fn.__name__ == "phase_latent_variable_model" and mp.Nb == 2 and mp.noisemodel == "NegativeBinomial"
    with pytest.raises(ValueError):
        P.preprocess_for_phase_estimation(ad, cyc, ph, Db, gene_selection_model="gmm")


def test_velocity_container_matches_reference():
    z, ad, cyc, ph = _inputs()
    Db = P.make_design_matrix(ad, ids="batch")
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1)
    spd = C.AngularSpeed.trivial_prior(["b0", "b1"], harmonics=1)
    mv = P.preprocess_for_velocity_estimation(ad, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=1,
                                              count_factor=mp.count_factor, ω_n_harmonics=1)
    for k in ("D", "Db", "ν", "μγ", "σγ", "μβ", "σβ", "μνω", "σνω", "μνg", "σνg", "φxy_prior", "count_factor", "S", "U",
              "logU", "σsgc"):
        got = getattr(mv, k).numpy()
        assert got.shape == z["vel_" + k].shape, (k, got.shape, z["vel_" + k].shape)
        assert np.allclose(got, z["vel_" + k], rtol=1e-6, atol=1e-6), k
    assert mv.model_type == "lrmn" == str(z["vel_model_type"])
    assert mv.guide_fn.__name__ == "velocity_latent_variable_guide_LRMN" and mv.Nhω == 3 and mv.Nx == 2
    from velocycle_amd.spec import spec_from_metaparams
    sp = spec_from_metaparams(mv, "velocity")
    assert (sp.guide, sp.Hw, sp.Nx, sp.Nb) == ("lrmn", 1, 2, 2) and sp.mu_nuw.shape == (2, 3)


def test_containers_roundtrip_and_label_quirk(tmp_path):
    cyc = C.Cycle.from_array(np.arange(6.0).reshape(3, 2), np.ones((3, 2)), ["g1", "g2"])
    assert list(cyc.means.index) == ["nu0", "nu1_cos", "nu1_sin"]        # labels swapped vs arithmetic
    cyc.save(tmp_path / "c.csv")
    back = C.Cycle.load(tmp_path / "c.csv")
    assert np.allclose(back.means.values, cyc.means.values) and np.allclose(back.stds.values, 1.0)
    assert back.harmonics == 1 and len(back) == 2 and back.genes == ["g1", "g2"]
    sp = C.AngularSpeed.trivial_prior(["a"], harmonics=0)
    assert sp.means.shape == (1, 1) and float(sp.stds.values[0, 0]) == 3.0
    ph = C.Phases.from_array(np.array([[1.0, 0.0], [0.0, -1.0]]), ["c1", "c2"])
    assert np.allclose(ph.phis.numpy(), [0.0, 1.5 * np.pi])
    ph.save(tmp_path / "p.csv")
    assert np.allclose(C.Phases.load(tmp_path / "p.csv").phi_xy.values, ph.phi_xy.values)
    with pytest.raises(AssertionError):
        C.Phases.from_array(np.zeros((3, 2)))


def test_host_eps_stream_equals_oracle_stream():
    from velocycle_amd.rng import draw_eps
    for case in ("phase_nb", "vel_mf_joint", "vel_lrmn_cond"):
        z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
        sp, p = H.spec_from_fixture(z), H.problem_from_fixture(z, torch.float32)
        a = draw_eps(sp, torch.Generator().manual_seed(3))
        b = orc.draw_eps(p, torch.Generator().manual_seed(3))
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (case, k)
        # the fixture's eps (second guide call of the reference run with the fixture's seed) is reproduced
        g = torch.Generator().manual_seed(int(z["seed"]))
        draw_eps(sp, g)
        second = draw_eps(sp, g)
        for k, v in z.items():
            if k.startswith("eps_"):
                assert np.array_equal(second[k[4:]].numpy(), v), (case, k)


def test_flat_clipped_adam_matches_oracle_restatement():
    from velocycle_amd.svi import FlatClippedAdam
    torch.manual_seed(0)
    args = {"lr": 0.03, "lrd": 0.97, "betas": (0.8, 0.99)}
    p0 = torch.randn(50)
    flat = FlatClippedAdam(50, args, "cpu")
    oa = orc.ClippedAdam(args)
    p, par = p0.clone(), {"x": p0.clone()}
    for i in range(7):
        g = torch.randn(50) * (30.0 if i == 3 else 1.0)      # exercises the +-10 clamp
        flat.step(p, g)
        par = oa.step(par, {"x": g})
    assert torch.allclose(p, par["x"], rtol=1e-6, atol=1e-7)


def test_pca_phase_prior_matches_reference():
    """Phases.from_pca_heuristic / max_corr / rotate against the reference's output on the same matrix."""
    z = H.load_fixture(f"{H.GOLDEN}/ref_phase_prior.npz")
    ad = AnnDataLite(z["S_sz"], z["S_sz"])
    ad.layers["S_sz"] = z["S_sz"]
    for tag, kw in (("a", dict(concentration=5.0, small_count=1)),
                    ("b", dict(concentration=1.0, small_count=0.1, zero_at_min_density=True, normalize_pcs=False))):
        p = C.Phases.from_pca_heuristic(ad, layer="S_sz", **kw)
        assert np.allclose(p.phi_xy.values, z["phixy_" + tag], atol=1e-6), tag
        shift, c, corr = p.max_corr(z["umis"], npoints=50)
        assert np.allclose([shift, c], z["maxcorr_" + tag], atol=1e-6) and np.allclose(corr, z["corr_" + tag], atol=1e-6)
        p.rotate(angle=-shift)
        assert np.allclose(p.phi_xy.values, z["rot_" + tag], atol=1e-6)
    with pytest.raises(ValueError):
        C.Phases.from_pca_heuristic(ad, layer="nope")


@pytest.mark.parametrize("normalize", [False, True])
def test_sparse_layers_give_the_same_containers_as_dense(normalize):
    """.h5ad files hold scipy sparse layers; the stand-in AnnData keeps them sparse through gene selection
    (`adata[:, genes].copy()`, preprocessing.py:20-63) and both preprocess_* produce the same tensors as from dense layers."""
    import scipy.sparse as sps
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.workloads import make_velocity_spec
    sp = make_velocity_spec(300, 40, "vjoint", seed=3)
    S, U = sp.S.t().numpy(), sp.U.t().numpy()
    res = []
    for sparse in (False, True):
        ad = AnnDataLite(sps.csr_matrix(S) if sparse else S, sps.csc_matrix(U) if sparse else U)
        genes = list(ad.var.index)[5:35]                     # the cycle prior knows a subset: intersection slices the layers
        cyc = C.Cycle.from_array(sp.mu_nu.T.numpy()[:, 5:35], sp.sd_nu.T.numpy()[:, 5:35], genes)
        ph = C.Phases.from_array(sp.phixy_prior.T.numpy(), cell_names=list(ad.obs.index))
        mp = P.preprocess_for_phase_estimation(ad, cyc, ph, torch.ones(300, 1), n_harmonics=1, normalize=normalize)
        spd = C.AngularSpeed.trivial_prior(["c0"], harmonics=1)
        mv = P.preprocess_for_velocity_estimation(ad, cyc, ph, spd, torch.ones(300, 1), torch.ones(300, 1), n_harmonics=1,
                                                  count_factor=mp.count_factor, normalize=normalize)
        res.append((mp, mv))
    for dense, sparse in zip(*res):
        assert dense.Ng == sparse.Ng == 30
        for f in dense._fields:
            x, y = getattr(dense, f), getattr(sparse, f)
            if isinstance(x, torch.Tensor):
                assert torch.equal(x, y), f


def test_dense_non_integer_layers_follow_the_reference_branches():
    """Dense ndarray layers with non-integer values: the reference's phase preprocessing keeps them as floats (its
    `.A` attempt fails for an ndarray -> `except` branch, preprocessing.py:141-147), its velocity preprocessing casts to
    int64 in both branches (:243-252).  Pinned by the reference's own output on such layers."""
    z, ad, cyc, ph = _inputs()
    ad2 = AnnDataLite(z["S"] * 0.5 + 0.25, z["U"] * 0.75)
    ad2.obs["batch"] = list(z["batch"])
    Db = P.make_design_matrix(ad2, ids="batch")
    mp2 = P.preprocess_for_phase_estimation(ad2, cyc, ph, Db, n_harmonics=1)
    for k in ("S", "U", "logS", "count_factor"):
        assert np.allclose(getattr(mp2, k).numpy(), z["nonint_phase_" + k], rtol=1e-6, atol=1e-6), k
    assert (mp2.S.numpy() != np.floor(mp2.S.numpy())).any()                 # the halves survived
    spd = C.AngularSpeed.trivial_prior(["b0", "b1"], harmonics=1)
    mv2 = P.preprocess_for_velocity_estimation(ad2, cyc, ph, spd, Db.float(), Db.float(), n_harmonics=1,
                                               count_factor=mp2.count_factor, ω_n_harmonics=1)
    for k in ("S", "U", "logU"):
        assert np.allclose(getattr(mv2, k).numpy(), z["nonint_vel_" + k], rtol=1e-6, atol=1e-6), k
    assert (mv2.S.numpy() == np.floor(mv2.S.numpy())).all()                 # truncated, as the reference does


def test_gene_selection_copies_only_the_layers_that_are_read():
    """adata[:, genes].copy() (preprocessing.py:20-63 filter_shared_genes): the stand-in selects a layer when it is first
    read, so the float64 logS / logU a phase preprocess left on the object are not copied for a velocity preprocess that
    never looks at them; values, order and independence from the parent are those of an eager copy."""
    rng = np.random.default_rng(0)
    S, U = rng.poisson(2.0, (50, 12)).astype(np.float32), rng.poisson(1.0, (50, 12)).astype(np.float32)
    ad = AnnDataLite(S, U)
    ad.layers["logS"] = np.log(S.astype(np.float64) + 1)
    genes = list(ad.var.index)
    keep = [genes[7], genes[2], genes[9]]
    sub = ad[:, keep].copy()
    assert list(sub.var.index) == keep and sub.shape == (50, 3)
    assert "logS" not in sub.layers._d and "logS" in sub.layers                        # present, not materialised
    assert np.array_equal(sub.layers["unspliced"], U[:, [7, 2, 9]]) and np.array_equal(sub.X, S[:, [7, 2, 9]])
    sub.layers["unspliced"][0, 0] = -1.0
    assert ad.layers["unspliced"][0, 7] == U[0, 7]                                     # a copy, not a view
    sub.layers["new"] = np.zeros((50, 3))
    assert set(sub.layers.keys()) == {"spliced", "unspliced", "logS", "new"} and len(sub.layers) == 4
    assert np.array_equal(sub.layers["logS"], ad.layers["logS"][:, [7, 2, 9]])
    again = sub.copy()
    assert again is sub                                                               # already owns its data
    rows = ad[[3, 1], keep]
    assert np.array_equal(rows.layers["spliced"], S[np.ix_([3, 1], [7, 2, 9])])


def test_lazy_layers_never_leak_a_placeholder_and_copy_keeps_them_lazy():
    """ADVICE r3 (anndata_lite.py): `adata.layers` is a MutableMapping, so dict(layers), {**layers}, copy(), pop(),
    setdefault() and items() all hand out arrays, never the `_LazyLayer` placeholder of an entry that was installed lazily;
    AnnDataLite.copy() carries such an entry over unbuilt (the memory saving the class exists for)."""
    from velocycle_amd.anndata_lite import _LazyLayer
    rng = np.random.default_rng(1)
    S, U = rng.poisson(2.0, (20, 6)).astype(np.float32), rng.poisson(1.0, (20, 6)).astype(np.float32)
    built = []

    def make():
        built.append(1)
        return np.log(S.astype(np.float64) + 1)
    ad = AnnDataLite(S, U)
    ad.layers.set_lazy("logS", make)
    cp = ad.copy()
    assert built == [] and isinstance(cp.layers._raw("logS"), _LazyLayer)          # copy(): still lazy on both sides
    assert np.array_equal(cp.layers["spliced"], S) and cp.layers["spliced"] is not ad.layers["spliced"]
    lc = ad.layers.copy()
    assert built == [] and isinstance(lc._raw("logS"), _LazyLayer)
    for view in (dict(ad.layers), {**ad.layers}, dict(ad.layers.items())):
        assert set(view) == {"spliced", "unspliced", "logS"}
        assert all(isinstance(v, np.ndarray) for v in view.values())
    assert len(built) == 1                                                          # built once, then cached
    assert isinstance(lc.setdefault("logS", None), np.ndarray) and isinstance(lc.pop("logS"), np.ndarray)
    assert "logS" not in lc and "logS" in ad.layers
    assert np.array_equal(cp.layers["logS"], ad.layers["logS"])
    sub = ad[:, list(ad.var.index)[:2]]
    assert all(isinstance(v, np.ndarray) for v in dict(sub.layers).values()) and sub.layers["logS"].shape == (20, 2)


def test_csr_side_channel_is_dropped_when_the_dense_counts_were_replaced_or_edited():
    """ADVICE r2 (engine.py:94): the CSR copy of a sparse layer may feed the engine only while it provably describes the
    same data as the dense S / U fields of the container, which users edit with `_replace` or in place."""
    import scipy.sparse as sp
    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.spec import spec_from_metaparams
    rng = np.random.default_rng(0)
    S = rng.poisson(1.0, size=(60, 12)).astype(np.float32)
    U = rng.poisson(0.5, size=(60, 12)).astype(np.float32)
    ad = AnnDataLite(sp.csr_matrix(S), sp.csr_matrix(U))
    cyc = C.Cycle.from_array(np.zeros((3, 12)), np.ones((3, 12)), list(ad.var.index))
    ph = C.Phases.from_array(np.ones((2, 60)), cell_names=list(ad.obs.index))
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, torch.ones(60, 1), n_harmonics=1, with_delta_nu=False)
    # sparse layers: the dense fields are placeholders until somebody reads them; the engine's spec does not
    assert isinstance(P.raw_field(mp, "S"), P._Lazy) and not P.raw_field(mp, "S").done
    sp0 = spec_from_metaparams(mp, "phase")
    assert sp0.S_csr is mp.S_csr and sp0.S is None and (sp0.Ng, sp0.Nc) == (12, 60)
    assert not P.raw_field(mp, "S").done and not P.raw_field(mp, "logU").done
    # reading them gives exactly what the dense-layer path builds
    mpd = P.preprocess_for_phase_estimation(AnnDataLite(S, U), cyc, ph, torch.ones(60, 1), n_harmonics=1, with_delta_nu=False)
    for name in ("S", "U", "logS", "logU", "count_factor"):
        a, b = getattr(mp, name), getattr(mpd, name)
        assert torch.equal(a, b) and a.stride() == b.stride() and a.dtype == b.dtype, name
    assert np.array_equal(ad.layers["logS"], AnnDataLite(S, U).layers.get("logS", ad.layers["logS"]))
    assert ad.layers["logS"].dtype == np.float64 and ad.layers["logS"].shape == (60, 12)
    assert mp.S_csr is not None and P.csr_is_current(mp.S_csr, mp.S) and P.csr_is_current(mp.S_csr, P.raw_field(mp, "S"))
    assert spec_from_metaparams(mp, "phase").S_csr is mp.S_csr
    # same-shape replacement (subsampled / permuted / normalised counts): the stale CSR must not be used
    mp2 = mp._replace(S=mp.S.flip(1).contiguous())
    assert spec_from_metaparams(mp2, "phase").S_csr is None
    # other fields replaced (what fit() itself does for count_factor): still current
    mp3 = mp._replace(count_factor=mp.count_factor * 0)
    assert spec_from_metaparams(mp3, "phase").S_csr is mp.S_csr
    # in-place edit of the dense tensor: version counter moves, CSR dropped
    mp.S.mul_(2.0)
    assert not P.csr_is_current(mp.S_csr, mp.S)
    assert spec_from_metaparams(mp, "phase").S_csr is None
