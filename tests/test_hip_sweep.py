"""GPU: sweep over the compiled kernel set (harmonics, batches, conditioning patterns, noise models, guides,
both genes-per-lane layouts) on small ragged problems, every gradient against the float64 oracle."""
import itertools

import numpy as np
import pytest
import torch

from oracle import velocycle_oracle as orc

pytestmark = pytest.mark.gpu


def _problem(kind, guide, noise, H, Hw, Nb, Nx, cond_sites, Nc, Ng, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    phi = torch.rand(Nc, generator=g, dtype=torch.float64) * 6.28
    Nh, Nhw = 2 * H + 1, 2 * Hw + 1
    # (generator passed to poisson too: without it the counts came from the unseeded global RNG -- another data set in every
    # process, and a tolerance that held on most of them)
    S = torch.poisson(torch.rand(Ng, Nc, generator=g, dtype=torch.float64) * 6, generator=g).double()
    U = torch.poisson(torch.rand(Ng, Nc, generator=g, dtype=torch.float64) * 2, generator=g).double()
    with_dnu = Nb > 0
    nb = max(Nb, 1)
    batch = torch.randint(0, nb, (Nc,), generator=g)
    Db = torch.stack([(batch == b).double() for b in range(nb)])
    cond = {}
    kw = dict(kind=kind, guide=guide, noisemodel=noise, with_delta_nu=with_dnu, H=H, S=S,
              count_factor=0.1 * r(Nc), Db=Db, mu_nu=torch.cat([1.0 + 0.3 * r(Ng, 1), 0.2 * r(Ng, Nh - 1)], 1),
              sd_nu=0.2 + 0.1 * torch.rand(Ng, Nh, generator=g, dtype=torch.float64),
              phixy_prior=2.0 * torch.stack([torch.cos(phi), torch.sin(phi)], 1),
              sd_dnu=(0.05 + 0.1 * torch.rand(nb, Ng, generator=g, dtype=torch.float64)) if kind == "phase" else 0.01,
              sigma_ln_s=0.5 if kind == "phase" else 0.1)
    if kind == "velocity":
        cb = torch.randint(0, Nx, (Nc,), generator=g)
        kw.update(U=U, D=torch.stack([(cb == x).double() for x in range(Nx)]), Hw=Hw,
                  mu_gamma=0.1 * r(Ng), sd_gamma=torch.full((Ng,), 0.5, dtype=torch.float64),
                  mu_beta=2.0 + 0.1 * r(Ng), sd_beta=torch.full((Ng,), 1.0, dtype=torch.float64),
                  mu_nuw=torch.cat([0.4 + 0.05 * r(Nx, 1), 0.02 * r(Nx, Nhw - 1)], 1),
                  sd_nuw=torch.full((Nx, Nhw), 0.1, dtype=torch.float64))
    p = orc.Problem(**kw)
    vals = {"ϕxy": p.phixy_prior + 0.1 * r(Nc, 2), "ν": p.mu_nu + 0.05 * r(Ng, Nh), "Δν": 0.01 * r(nb, Ng),
            "shape_inv": 0.2 + torch.rand(Ng, generator=g, dtype=torch.float64),
            "logγg": 0.1 * r(Ng), "logβg": 2 + 0.1 * r(Ng), "νω": 0.4 + 0.01 * r(max(Nx, 1), Nhw), "rho_real": 3 + 0.2 * r(Ng)}
    p.condition_on = {s: vals[s] for s in cond_sites}
    return p


def _check(p, gpl=None, tuning=None):
    from tests.helpers import spec_from_problem
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.tuning import Tuning
    tuning = (tuning or Tuning()).replace(genes_per_lane=gpl or 0)
    gen = torch.Generator().manual_seed(1)
    first = orc.draw_eps(p, gen)
    eps = orc.draw_eps(p, gen)
    par = orc.init_params(p, first.get("_cov_factor_draw"))
    for k in par:                                   # move off the prior means so that every term is exercised
        if torch.isfinite(par[k]).all():
            par[k] = par[k] + 0.05 * torch.randn(par[k].shape, generator=gen, dtype=torch.float64)
    eng = HipEngine(spec_from_problem(p), tuning=tuning)
    eng.set_params({k: v.float() for k, v in par.items()})
    eng.elbo_grad(eps=eng.pack_eps({k: v.float() for k, v in eps.items() if not k.startswith("_")}))
    torch.cuda.synchronize()
    par32 = {k: v.float().double() for k, v in par.items()}
    eps32 = {k: v.float().double() for k, v in eps.items() if not k.startswith("_")}
    l64, g64, _, _ = orc.loss_and_grads(p, par32, eps32)
    assert abs(eng.loss() - l64) <= 2e-5 * abs(l64) + 1e-3, (eng.loss(), l64)
    for name, got in eng.named(eng.grad).items():
        want = g64[name].numpy()
        fin = np.isfinite(want)
        err = np.abs(got.cpu().numpy()[fin] - want[fin]).max() if fin.any() else 0.0
        assert err <= 3e-3 * max(np.abs(want[fin]).max() if fin.any() else 0, 1e-2), (name, err)
    kind = eng.stats["main_kernel"]
    # batch offsets: a one-hot design matrix is folded per workgroup (the kernel's NB is 0 for any number of batches) unless the
    # tuning asks for the dense contraction; anything else keeps NB = Nb (or the run-time-sized set beyond 4 batches)
    if p.with_delta_nu and not eng.stats["generic"]:
        onehot = bool(((p.Db == 0) | (p.Db == 1)).all() and (p.Db.sum(0) == 1).all()) and not tuning.dense_batches
        assert eng.stats["onehot_batches"] == (p.Nb if onehot else 0), eng.stats
        assert kind.startswith(f"vc_main_kernel<{p.H},{0 if onehot else p.Nb},"), kind
    eng.close()
    return kind


def _dense(dense):
    from velocycle_amd.tuning import Tuning
    return Tuning(dense_batches=True) if dense else None


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("H,Nb", list(itertools.product((1, 2, 3), (0, 1, 2, 3, 4))))
def test_phase_kernels(H, Nb, dense):
    """dense = True: the compiled NB = 1..4 instantiations (batch offsets as a dense contraction per cell: what a design matrix
    that is not one-hot runs); False: one-hot batches folded per workgroup, in the shuffled cell order `_problem` draws."""
    if dense and Nb == 0:
        pytest.skip("no batches")
    for noise, gpl in (("NegativeBinomial", 8), ("Poisson", 4), ("Lognormal", None)):
        p = _problem("phase", "meanfield", noise, H, 0, Nb, 0, [], Nc=70 + 13 * H, Ng=9 + Nb, seed=H * 10 + Nb)
        _check(p, gpl, _dense(dense))


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("H,Nb", list(itertools.product((1, 2, 3), (0, 1, 2, 4))))
def test_velocity_joint_kernels(H, Nb, dense):
    if dense and Nb == 0:
        pytest.skip("no batches")
    for noise, guide, Hw, Nx, gpl in (("NegativeBinomial", "meanfield", 1, 2, 8), ("Poisson", "lrmn", 0, 1, 4),
                                      ("Lognormal", "meanfield", 2, 3, None)):
        p = _problem("velocity", guide, noise, H, Hw, Nb, Nx, [], Nc=90 + 7 * Nb, Ng=7 + H, seed=100 + H * 10 + Nb)
        k = _check(p, gpl, _dense(dense))
        assert "vfull" in k


@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("H,Nb", list(itertools.product((1, 2, 3), (0, 2, 3))))
def test_velocity_conditioned_kernels(H, Nb, dense):
    if dense and Nb == 0:
        pytest.skip("no batches")
    full = ["ϕxy", "ν", "shape_inv"] + (["Δν"] if Nb else [])
    for noise, guide, Hw, gpl in (("NegativeBinomial", "lrmn", 1, 8), ("NegativeBinomial", "meanfield", 3, 4),
                                  ("Poisson", "lrmn", 0, None), ("Lognormal", "meanfield", 1, None)):
        sites = [s for s in full if not (s == "shape_inv" and noise != "NegativeBinomial")]
        p = _problem("velocity", guide, noise, H, Hw, Nb, 2, sites, Nc=130, Ng=11, seed=200 + H * 10 + Nb)
        k = _check(p, gpl, _dense(dense))
        assert "vu_" in k          # S term hoisted


@pytest.mark.parametrize("sites", [["ϕxy"], ["ν"], ["shape_inv"], ["ν", "ϕxy"], ["logγg"], ["logβg", "νω"],
                                   ["rho_real"], ["Δν", "logγg", "logβg", "νω", "rho_real"]])
def test_partial_conditioning(sites):
    for guide in ("meanfield", "lrmn"):
        s = [x for x in sites if not (x == "rho_real" and guide != "lrmn")]
        p = _problem("velocity", guide, "NegativeBinomial", 1, 1, 2, 2, s, Nc=101, Ng=10, seed=7)
        k = _check(p, None)
        assert "vfull" in k


@pytest.mark.parametrize("Nc,Ng", [(1, 1), (3, 300), (65, 2), (64, 256), (257, 513), (700, 5)])
def test_ragged_and_tiny_shapes(Nc, Ng):
    for gpl in (4, 8):
        p = _problem("velocity", "meanfield", "NegativeBinomial", 1, 1, 1, 1, [], Nc=Nc, Ng=Ng, seed=Nc + Ng)
        _check(p, gpl)


@pytest.mark.parametrize("rank,Nx,Hw", [(1, 1, 0), (3, 2, 1), (8, 3, 2), (8, 9, 3)])
def test_lrmn_ranks_and_many_speed_coefficients(rank, Nx, Hw):
    """LRMN guide with rank 1 / 3 / 8 (the compiled maximum) and up to Nx (2 Hw + 1) = 63 angular-speed coefficients:
    the loops over the low-rank factors and over the nu_omega outputs in K_pre, K_post and K_fin."""
    for sites in ([], ["ϕxy", "ν", "shape_inv"]):
        p = _problem("velocity", "lrmn", "NegativeBinomial", 1, Hw, 0, Nx, sites, Nc=150, Ng=70, seed=300 + rank + Nx)
        p.rho_rank = rank
        _check(p, None)


@pytest.mark.parametrize("Ng", [1, 3, 65, 66, 129])
def test_lrmn_with_a_nearly_empty_last_gene_block(Ng):
    """Fewer real genes in the last 64-gene block than low-rank factors: the eps_W broadcast of K_pre must not depend
    on lanes that belong to padded genes (a cross-lane read after divergence did, caught at rank >= 7 with Ng = 70)."""
    p = _problem("velocity", "lrmn", "NegativeBinomial", 1, 1, 0, 2, [], Nc=90, Ng=Ng, seed=400 + Ng)
    _check(p, None)


# ---- outside the compiled fast set: the run-time-sized kernel set (csrc/vc_generic_kernels.hip) -------------------------------
# The reference takes any number of harmonics (utils.py:400-437), any number of batches (preprocessing.py:65-93: one design
# column per unique id; phase_inference_model.py:374-377, velocity_inference_model.py:360) and any rho_rank (preprocessing.py:239,
# velocity_inference_guide.py:91-92); round 3 refused H > 3, > 4 batches, rank > 8 with VC_ERR_UNSUPPORTED.

@pytest.mark.parametrize("H,Nb", [(4, 0), (5, 2), (4, 5), (2, 8), (7, 9)])
def test_generic_phase_kernels(H, Nb):
    for noise in ("NegativeBinomial", "Poisson", "Lognormal"):
        p = _problem("phase", "meanfield", noise, H, 0, Nb, 0, [], Nc=70 + 13 * H, Ng=9 + Nb, seed=H * 10 + Nb)
        k = _check(p, None, _dense(True))          # (a one-hot design with H <= 3 now runs on the fast set: test_onehot_*)
        assert "generic_phase" in k and "gpl2" in k, k


@pytest.mark.parametrize("H,Nb", [(4, 0), (5, 5), (1, 8), (3, 6)])
def test_generic_velocity_joint_kernels(H, Nb):
    for noise, guide, Hw, Nx in (("NegativeBinomial", "meanfield", 1, 2), ("Poisson", "lrmn", 0, 1), ("Lognormal", "meanfield", 4, 3),
                                 ("NegativeBinomial", "lrmn", 2, 2)):
        p = _problem("velocity", guide, noise, H, Hw, Nb, Nx, [], Nc=90 + 7 * Nb, Ng=7 + H, seed=100 + H * 10 + Nb)
        k = _check(p, None, _dense(True) if Hw <= 3 else None)      # (Hw = 4 is generic by itself: there with the one-hot design as it is)
        assert "generic_vfull" in k, k


@pytest.mark.parametrize("H,Nb", [(4, 0), (2, 5), (5, 8)])
def test_generic_velocity_conditioned_kernels(H, Nb):
    full = ["ϕxy", "ν", "shape_inv"] + (["Δν"] if Nb else [])
    for noise, guide, Hw in (("NegativeBinomial", "lrmn", 1), ("NegativeBinomial", "meanfield", 5), ("Poisson", "lrmn", 0),
                             ("Lognormal", "meanfield", 1)):
        sites = [s for s in full if not (s == "shape_inv" and noise != "NegativeBinomial")]
        p = _problem("velocity", guide, noise, H, Hw, Nb, 2, sites, Nc=130, Ng=11, seed=200 + H * 10 + Nb)
        k = _check(p, None, _dense(True) if Hw <= 3 else None)
        assert "generic_vu" in k, k          # S term hoisted by the generic S-only kernel


@pytest.mark.parametrize("rank,Nx,Hw", [(12, 2, 1), (9, 1, 0), (16, 3, 2), (5, 10, 3), (12, 13, 3)])
def test_generic_lrmn_rank_and_speed_coefficients(rank, Nx, Hw):
    """LRMN rank above the compiled 8 (the verdict's rank 12), and more than 64 angular-speed coefficients (10 x 7 = 70, 13 x 7 = 91)."""
    for sites in ([], ["ϕxy", "ν", "shape_inv"]):
        p = _problem("velocity", "lrmn", "NegativeBinomial", 1, Hw, 0, Nx, sites, Nc=150, Ng=70, seed=300 + rank + Nx)
        p.rho_rank = rank
        k = _check(p, None)
        assert "generic" in k, k


@pytest.mark.parametrize("Nc,Ng", [(1, 1), (3, 300), (65, 2), (257, 513)])
def test_generic_set_on_fast_set_sizes_equals_the_fast_set(Nc, Ng):
    """Tuning(force_generic=True): the run-time-sized kernels on a configuration the fast set covers -- same oracle, same tolerances,
    ragged and tiny shapes -- and the two kernel sets against each other."""
    from tests.helpers import spec_from_problem
    from velocycle_amd.engine import HipEngine
    p = _problem("velocity", "lrmn", "NegativeBinomial", 2, 1, 2, 2, [], Nc=Nc, Ng=Ng, seed=Nc + Ng)
    from velocycle_amd.tuning import Tuning
    fast = _check(p, None)
    gen = _check(p, None, Tuning(force_generic=True))
    assert "generic" in gen and "generic" not in fast


def test_the_fast_instantiations_stay_selected_where_they_exist():
    from tests.helpers import spec_from_problem
    from velocycle_amd.engine import HipEngine
    for H, Nb, rank in ((3, 4, 8), (1, 0, 5)):
        p = _problem("velocity", "lrmn", "NegativeBinomial", H, 3, Nb, 3, [], Nc=90, Ng=12, seed=5)
        p.rho_rank = rank
        from velocycle_amd.tuning import Tuning
        e = HipEngine(spec_from_problem(p))          # one-hot batches: folded, the kernel without batch terms
        assert not e.stats["generic"] and e.stats["main_kernel"].startswith(f"vc_main_kernel<{H},0,vfull_nb,gpl"), e.stats
        assert e.stats["onehot_batches"] == Nb
        e.close()
        e = HipEngine(spec_from_problem(p), tuning=Tuning(dense_batches=True))
        assert not e.stats["generic"] and e.stats["main_kernel"].startswith(f"vc_main_kernel<{H},{Nb},vfull_nb,gpl"), e.stats
        e.close()


def test_generic_set_runs_whole_fits():
    """SVIRunner on a generic engine: perf mode picks the unfused kernel sequence by itself (the fused steps exist for the fast
    set only and say so), parity mode follows the float64 oracle's trajectory."""
    from tests.helpers import spec_from_problem
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    p = _problem("velocity", "lrmn", "NegativeBinomial", 4, 1, 5, 2, [], Nc=300, Ng=40, seed=11)
    p.rho_rank = 10
    opt = {"lr": 0.03, "lrd": 0.99, "betas": (0.8, 0.99)}
    e = HipEngine(spec_from_problem(p))
    assert e.stats["generic"] and e.stats["launches_per_step"] == 3
    r = SVIRunner(e, opt, mode="perf", seed=3)
    assert r.adam_impl == "hip"
    r.run_perf(30)
    l = np.array(r.perf_losses())
    assert np.isfinite(l).all() and l[-5:].mean() < l[:5].mean() and e.status()[0]
    with pytest.raises(NotImplementedError, match="unfused"):
        SVIRunner(e, opt, mode="perf", seed=3, adam_impl="fused3").run_perf(1)
    e.close()
    e = HipEngine(spec_from_problem(p))
    run = SVIRunner(e, opt, mode="parity", seed=11)
    losses = np.array([run.step() for _ in range(8)])
    l64, _ = orc.fit(p, opt, 8, seed=11)
    assert np.allclose(losses[:3], np.array(l64)[:3], rtol=2e-5) and np.allclose(losses, np.array(l64), rtol=5e-4), (losses, l64)
    e.close()


def test_unsupported_configurations_raise():
    from tests.helpers import spec_from_problem
    from velocycle_amd.engine import HipEngine
    p = _problem("phase", "meanfield", "NegativeBinomial", 1, 0, 1, 0, [], Nc=20, Ng=5, seed=1)
    sp = spec_from_problem(p)
    sp.H = 80                                   # 2 H + 1 = 161 rows of per-gene state per wave: beyond the LDS
    sp.mu_nu = torch.zeros(5, 161)
    sp.sd_nu = torch.ones(5, 161)
    with pytest.raises(NotImplementedError, match="LDS"):
        HipEngine(sp)
    sp2 = spec_from_problem(p)
    sp2.noisemodel = "Gaussian"
    with pytest.raises(ValueError, match="not allowed"):
        HipEngine(sp2)
    sp3 = spec_from_problem(p)
    sp3.sd_nu = sp3.sd_nu * 0 - 1
    with pytest.raises(ValueError):
        HipEngine(sp3)
    sp4 = spec_from_problem(p)
    sp4.condition_on = {"nonsense": torch.zeros(3)}
    with pytest.raises(ValueError):
        HipEngine(sp4)


# ---- one-hot batch designs: any number of batches on the fast kernel set, at no cost per cell ---------------
# The reference's make_design_matrix builds one indicator column per unique id (preprocessing.py:65-93), used as
# einsum("bgc,bgc->gc", Db, dnu) (phase_inference_model.py:374-377) / its velocity form (velocity_inference_model.py:360).

@pytest.mark.parametrize("order", ["contiguous", "shuffled"])
@pytest.mark.parametrize("Nb", [2, 5, 8, 12])
def test_onehot_batches_any_number_on_the_fast_set(Nb, order):
    """Nb in {2, 5, 8, 12} batches, contiguous (anndata.concat(..., label="batch")) and interleaved (cells re-ordered by batch inside
    the engine, results in the caller's order): phase, velocity joint (delta nu learned: its gradient is a sum over the batch's
    workgroups) and the tutorials' conditioned velocity stage, against the float64 oracle; the kernel asserted NB = 0."""
    for kind, guide, noise, sites, H in (("phase", "meanfield", "NegativeBinomial", [], 2),
                                         ("velocity", "meanfield", "NegativeBinomial", [], 1),
                                         ("velocity", "lrmn", "Poisson", [], 1),
                                         ("velocity", "lrmn", "NegativeBinomial", ["ϕxy", "ν", "Δν", "shape_inv"], 1)):
        p = _problem(kind, guide, noise, H, 1, Nb, 2, sites, Nc=400 + 31 * Nb, Ng=70, seed=500 + Nb)
        if order == "contiguous":
            idx = torch.argsort(p.Db.argmax(0), stable=True)
            p.Db = p.Db[:, idx].contiguous()
        for gpl in (4, 8):
            k = _check(p, gpl)
            assert k.startswith(f"vc_main_kernel<{H},0,") and "generic" not in k, k


def test_a_design_matrix_that_is_not_onehot_keeps_the_dense_contraction():
    """Soft batch memberships (columns 0.3 / 0.7) and a cell that belongs to two batches: the reference's einsum takes any matrix;
    the engine then runs the dense NB = Nb kernels (<= 4 batches) or the run-time-sized set (more) -- same oracle, same bars."""
    for Nb, soft in ((2, True), (3, False), (6, True)):
        p = _problem("velocity", "meanfield", "NegativeBinomial", 1, 1, Nb, 2, [], Nc=300, Ng=40, seed=600 + Nb)
        if soft:
            p.Db = 0.3 * p.Db + 0.7 * p.Db.roll(1, 0)
        else:
            p.Db[1, 5] = 1.0
            p.Db[0, 5] = 1.0
        k = _check(p, None)
        assert ("generic" in k) == (Nb > 4), k
        if Nb <= 4:
            assert k.startswith(f"vc_main_kernel<1,{Nb},"), k


def test_onehot_batches_on_ragged_tilings_and_an_empty_batch():
    """A batch without cells on this rank, a batch of three cells, tiles forced ragged: every workgroup still lies inside one batch
    (vc_host_logic.h: vc_tile_batches) and every cell is evaluated exactly once."""
    from velocycle_amd.tuning import Tuning
    p = _problem("velocity", "meanfield", "NegativeBinomial", 1, 1, 5, 2, [], Nc=1500, Ng=300, seed=77)
    b = torch.zeros(1500, dtype=torch.long)
    b[700:703] = 3
    b[703:] = 4
    b[100:400] = 2                                   # batch 1 is empty
    p.Db = torch.stack([(b == q).double() for q in range(5)])
    for cw in (0, 7, 29):
        k = _check(p, None, Tuning(cells_per_wave=cw))
        assert k.startswith("vc_main_kernel<1,0,"), k

