"""GPU: the three-launch fused step (vc_svi_step_fused: K_main -> K_tail -> K_omega, SVIRunner adam_impl="fused3") against
the unfused kernel sequence it replaces (K_pre -> K_main -> K_post -> K_fin -> ClippedAdam, adam_impl="hip" / "fused"):
same Philox stream, same arithmetic statement by statement, so the trajectories must coincide -- parameters and
optimiser moments to float32 rounding of a few reassociated sums (observed: bit-identical in most configurations), losses
to 2e-7 relative (the float32 prior / guide terms are grouped per role instead of per thread before they enter the
fp64 loss assembly).  Every model / guide / noise /
conditioning combination of the step fixtures, plus medium sizes with ragged tiles and several gene blocks."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
OPT = {"lr": 0.03, "lrd": 0.995, "betas": (0.8, 0.99)}


def _run(spec, impl, n, use_graph, seed=7, tuning=None):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    e = HipEngine(spec, tuning=tuning)
    r = SVIRunner(e, OPT, mode="perf", seed=seed, use_graph=use_graph, adam_impl=impl)
    r.run_perf(n)
    out = dict(p=e.params.clone().cpu(), l=np.array(r.perf_losses()), m=r.opt.m.clone().cpu(), v=r.opt.v.clone().cpu(),
               g=e.grad.clone().cpu(), sd=int(r.step_dev.item()), status=e.status())
    e.close()
    return out


def _same(a, b, what, rtol=2e-5, atol=2e-6):
    a, b = a.double().numpy(), b.double().numpy()
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), what
    assert np.allclose(a[fin], b[fin], rtol=rtol, atol=atol), (what, np.abs(a[fin] - b[fin]).max())


@pytest.mark.parametrize("case", H.STEP_CASES)
def test_fused_step_equals_unfused_sequence(case):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    n = 15
    ref = _run(spec, "hip", n, False)
    for use_graph in (False, True):
        got = _run(spec, "fused3", n, use_graph)
        assert got["sd"] == n and ref["sd"] == n and got["status"][0]
        assert len(got["l"]) == n and np.allclose(got["l"], ref["l"], rtol=2e-7, atol=0), np.abs(got["l"] / ref["l"] - 1).max()
        _same(got["p"], ref["p"], f"{case}: params")
        _same(got["m"], ref["m"], f"{case}: exp_avg")
        _same(got["v"], ref["v"], f"{case}: exp_avg_sq", rtol=1e-4)
        _same(got["g"][4:], ref["g"][4:], f"{case}: last gradient", rtol=1e-4, atol=1e-4)
    # graph replay == eager launches of the fused step, bit for bit
    a, b = _run(spec, "fused3", n, True), _run(spec, "fused3", n, False)
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    assert torch.equal(nz(a["p"]), nz(b["p"])) and np.array_equal(a["l"], b["l"])


@pytest.mark.parametrize("mode,ncond", [("vcond", 1), ("vcond", 2), ("vcond_mf", 1)])
def test_tutorial_flow_two_launch_step_equals_the_three_launch_step(mode, ncond):
    """Tutorial flow on one rank (U-only kernel, phases / nu / shape_inv conditioned): from the third step of a run on, K_tail's
    gene blocks and K_omega's blocks go out as ONE launch (vc_launch_tail_merged) and K_tail's cell blocks not at all -- the
    partials of d loglik / d nu_omega come from K_main (pw_inline).  The launch structure changes nothing: bit-identical
    parameters, moments and losses after 25 steps (Tuning(tail_merged=False): three launches).  Against the cell blocks' own partial
    sums (Tuning(pw_inline="off")) only the association of one sum over the cells differs: equal to float32 rounding after 2 steps."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3001, 300, mode, n_conditions=ncond, Hw=1, seed=5)
    e = HipEngine(spec)
    assert e.stats["main_kernel"].startswith("vc_main_kernel<1,") and "vu_" in e.stats["main_kernel"]
    e.close()
    from velocycle_amd.tuning import Tuning
    two = _run(spec, "fused3", 25, False)
    three = _run(spec, "fused3", 25, False, tuning=Tuning(tail_merged=False))
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    assert np.array_equal(two["l"], three["l"]) and torch.equal(nz(two["p"]), nz(three["p"]))
    assert torch.equal(nz(two["m"]), nz(three["m"])) and torch.equal(nz(two["v"]), nz(three["v"]))
    assert two["sd"] == 25 and two["status"][0]
    a = _run(spec, "fused3", 2, False)
    b = _run(spec, "fused3", 2, False, tuning=Tuning(pw_inline="off"))
    assert np.allclose(a["l"], b["l"], rtol=2e-7, atol=0)
    _same(a["p"], b["p"], "params after 2 steps", rtol=2e-6, atol=2e-7)
    _same(a["g"][4:], b["g"][4:], "gradient of step 2", rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("mode,ncond,cw", [("vjoint", 1, "37"), ("vcond", 2, None), ("vjoint", 2, "29"), ("vcond_mf", 1, None)])
def test_fused_step_medium_sizes(mode, ncond, cw):
    """3001 (x n_conditions) cells x 300 genes: several gene blocks, many cell blocks, ragged tails, Nx = Nb = 2."""
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    tun = Tuning(cells_per_wave=int(cw)) if cw else None
    spec = make_velocity_spec(3001, 300, mode, n_conditions=ncond, Hw=1, seed=5)
    # two steps: nothing but reassociation can differ yet -> tight; twelve steps: the optimiser has amplified that rounding
    # where gradients pass through zero (same yardstick as the float32-vs-float64 trajectory tests) -> loose
    ref, got = _run(spec, "hip", 2, False, tuning=tun), _run(spec, "fused3", 2, False, tuning=tun)
    assert np.allclose(got["l"], ref["l"], rtol=2e-7, atol=0), np.abs(got["l"] / ref["l"] - 1).max()
    _same(got["p"], ref["p"], "params after 2 steps", rtol=2e-6, atol=2e-7)
    _same(got["m"], ref["m"], "exp_avg after 2 steps", rtol=1e-5, atol=1e-6)
    _same(got["g"][4:], ref["g"][4:], "gradient of step 2", rtol=1e-5, atol=1e-5)
    ref, got = _run(spec, "hip", 12, False, tuning=tun), _run(spec, "fused3", 12, True, tuning=tun)
    assert np.allclose(got["l"], ref["l"], rtol=1e-6, atol=0), np.abs(got["l"] / ref["l"] - 1).max()
    _same(got["p"], ref["p"], "params", rtol=1e-3, atol=1e-4)
    _same(got["m"], ref["m"], "exp_avg", rtol=2e-3, atol=2e-3)


def test_fused_phase_medium_and_resume_mid_run():
    """phase model at 3000 x 200, and: a run interrupted by state_dict()/load_state_dict() into a new engine continues
    the same trajectory (the next step's sample is re-drawn from the restored parameters)."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_phase_spec
    spec = make_phase_spec(3000, 200, seed=5)
    ref, got = _run(spec, "hip", 12, False), _run(spec, "fused3", 12, True)
    assert np.allclose(got["l"], ref["l"], rtol=1e-6, atol=0)
    _same(got["p"], ref["p"], "params", rtol=1e-3, atol=1e-4)
    e1 = HipEngine(spec)
    r1 = SVIRunner(e1, OPT, mode="perf", seed=7)
    assert r1.adam_impl == "fused3"
    r1.run_perf(5)
    sd = r1.state_dict()
    e2 = HipEngine(spec)
    r2 = SVIRunner(e2, OPT, mode="perf", seed=7)
    r2.load_state_dict(sd)
    r2.run_perf(7)
    assert torch.equal(e2.params.cpu(), got["p"]) and np.array_equal(np.array(r2.perf_losses()), got["l"])
    e1.close()
    e2.close()


def test_fused_long_run_tracks_the_unfused_sequence():
    """600 steps at 3000 x 200 (eps ring wraps 200 times, the loss ring is used past its first thousand slots in the second
    half of the test): the two sequences start bit-close and end statistically indistinguishable -- every loss finite, no
    NaN / Inf latched, the mean of the last 100 losses within 2e-3, the fitted gene-level means correlated to 0.999; then a 1500-step continuation of the fused run stays finite and keeps descending."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3000, 200, "vjoint", n_conditions=1, Hw=1, seed=8)
    opt = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / 2000), "betas": (0.8, 0.99)}
    out = {}
    for impl in ("hip", "fused3"):
        e = HipEngine(spec)
        r = SVIRunner(e, opt, mode="perf", seed=21, use_graph=False, adam_impl=impl)
        r.run_perf(600)
        l = np.array(r.perf_losses())
        assert len(l) == 600 and np.isfinite(l).all() and e.status() == (True, -1, 0)
        out[impl] = (l, {k: v.detach().cpu().numpy().copy() for k, v in e.named().items()}, e, r)
    la, lb = out["hip"][0], out["fused3"][0]
    assert np.allclose(la[:5], lb[:5], rtol=1e-6)
    assert abs(la[-100:].mean() - lb[-100:].mean()) <= 2e-3 * abs(la[-100:].mean())
    assert lb[-100:].mean() < lb[:100].mean()
    for k in ("ν_locs", "logγg_locs", "logβg_locs"):
        a, b = out["hip"][1][k], out["fused3"][1][k]
        # element by element the two float32 Adam trajectories have drifted apart by now (same yardstick as the float32 /
        # float64 oracle runs); as fits they are the same: correlated to 0.999, typical distance a percent of the spread
        # (log gamma / log beta are the weakly identified ones: 0.99)
        cmin = 0.999 if k == "ν_locs" else 0.99
        assert np.corrcoef(a.ravel(), b.ravel())[0, 1] > cmin, (k, np.corrcoef(a.ravel(), b.ravel())[0, 1])
        assert np.median(np.abs(a - b)) <= 0.02 * max(a.std(), 1e-3) + 0.01, (k, np.median(np.abs(a - b)), a.std())
    e, r = out["fused3"][2], out["fused3"][3]
    r.run_perf(1500)
    l2 = np.array(r.perf_losses())
    assert len(l2) == 2100 and np.isfinite(l2).all() and e.status()[0]
    assert l2[-200:].mean() < l2[400:600].mean()
    for v in out.values():
        v[2].close()


@pytest.mark.parametrize("H,Nb", [(1, 0), (2, 1), (3, 2), (2, 4), (3, 0)])
def test_fused_step_sweep_over_the_kernel_set(H, Nb):
    """Fused vs unfused on small ragged problems over harmonics / batches / guides / noise models / conditioning patterns
    (the generator of tests/test_hip_sweep.py): 6 steps each; identical samples -> parameters to float32 rounding."""
    from tests.helpers import spec_from_problem
    from tests.test_hip_sweep import _problem
    cases = [("phase", "meanfield", "NegativeBinomial", 0, 0, []),
             ("phase", "meanfield", "Poisson", 0, 0, []),
             ("velocity", "meanfield", "NegativeBinomial", 1, 2, []),
             ("velocity", "lrmn", "NegativeBinomial", min(H, 2), 2, []),
             ("velocity", "lrmn", "NegativeBinomial", 0, 1, ["ϕxy", "ν", "shape_inv"] + (["Δν"] if Nb else [])),
             ("velocity", "meanfield", "Lognormal", 1, 1, ["νω"]),
             ("velocity", "meanfield", "Poisson", 2, 3, ["logγg"])]
    for i, (kind, guide, noise, Hw, Nx, cond) in enumerate(cases):
        p = _problem(kind, guide, noise, H, Hw, Nb, Nx, cond, Nc=150 + 31 * i, Ng=70 + 3 * i, seed=100 * H + 10 * Nb + i)
        spec = spec_from_problem(p)
        ref, got = _run(spec, "hip", 6, False, seed=3), _run(spec, "fused3", 6, False, seed=3)
        tag = f"H={H} Nb={Nb} case {i} {kind}/{guide}/{noise}/cond={cond}"
        assert got["status"][0] and np.allclose(got["l"], ref["l"], rtol=2e-6, atol=0), (tag, got["l"], ref["l"])
        _same(got["p"], ref["p"], tag + ": params", rtol=2e-4, atol=2e-5)
        # (exp_avg holds 0.2 x the LAST gradient: on these random small problems single genes sit on the relu kink of ElogU,
        # where a 1e-5 difference of the parameters moves their gradient by per cent)
        _same(got["m"], ref["m"], tag + ": exp_avg", rtol=2e-3, atol=5e-3)


@pytest.mark.parametrize("mode", ["vjoint", "vcond"])
def test_step_with_loss_equals_the_device_ring(mode):
    """SVIRunner.step_with_loss: the loss of every step reaches the host through the pinned ring the device writes into
    (no synchronise, no copy) -- the same steps, the same losses as run_perf, bit for bit, and the two can be mixed."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3000, 256, mode, 1, 1, seed=4)
    ea, eb = HipEngine(spec), HipEngine(spec)
    a = SVIRunner(ea, OPT, mode="perf", seed=11)
    b = SVIRunner(eb, OPT, mode="perf", seed=11)
    a.run_perf(40)
    la = a.perf_losses()
    b.run_perf(5)                                   # the ring starts on the device ...
    lb = b.perf_losses()
    lb += [b.step_with_loss() for _ in range(25)]   # ... moves to pinned host memory ...
    assert not b.loss_hist.is_cuda and b.loss_hist.is_pinned()
    b.run_perf(10)                                  # ... and run_perf keeps filling it
    lb += b.perf_losses()[30:]
    assert len(lb) == 40 and lb == la
    assert b.perf_losses() == la
    assert torch.equal(ea.params, eb.params)
    ea.close(); eb.close()


@pytest.mark.parametrize("mode", ["vjoint", "vcond"])
def test_fused_step_with_large_shard_cell_blocks(mode):
    """K_tail's cell blocks take 1024 instead of 256 cells on shards above 160 000 cells (all 16 waves of the block busy);
    forced here at a small size: same steps as the unfused sequence."""
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(2600, 200, mode, 2, 1, seed=9)
    a = _run(spec, "fused3", 6, False, tuning=Tuning(tail_cells=1024))
    b = _run(spec, "fused", 6, False)
    assert a["sd"] == b["sd"] == 6 and a["status"][0] and b["status"][0]
    assert np.allclose(a["l"], b["l"], rtol=2e-7, atol=0)
    _same(a["p"], b["p"], "params", rtol=2e-4, atol=2e-5)


def _bits_equal(a, b, what):
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    assert np.array_equal(a["l"], b["l"]), (what, "losses", np.abs(a["l"] / b["l"] - 1).max())
    for k in ("p", "m", "v"):
        assert torch.equal(nz(a[k]), nz(b[k])), (what, k, (nz(a[k]) - nz(b[k])).abs().max())
    assert torch.equal(nz(a["g"][4:]), nz(b["g"][4:])), (what, "g")
    assert a["sd"] == b["sd"] and a["status"][0] and b["status"][0]


@pytest.mark.parametrize("case", H.STEP_CASES)
def test_two_launch_step_equals_three_launch_step_on_the_fixtures(case):
    """Round 4: every single-rank step in TWO launches (vc_launch_tail2: K_tail's gene blocks, its cell blocks with the nu_omega
    chain inside on K_main's own partials, the loss block, the histogram blocks re-deriving the shape_inv update, the eps
    blocks -- side by side).  The launch structure changes nothing: parameters, moments, gradients AND losses bit for bit equal
    to the three-launch step (Tuning(tail2=False)) after 15 steps, on every model / guide / noise / conditioning of the step fixtures."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.tuning import Tuning
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    # like with like: the dense histogram tables are the default only where the one-launch tail runs; here both launch
    # structures evaluate them (the four-wave and the sixteen-wave blocks add the same slices in the same order)
    t2 = Tuning(hist_dense="dense")
    t3 = t2.replace(tail2=False, tail_merged=False)
    e = HipEngine(spec, tuning=t2)
    assert e.stats["launches_per_step"] == 2, e.stats          # every fixture is small enough for K_main's own partials
    e.close()
    two = _run(spec, "fused3", 15, False, tuning=t2)
    e = HipEngine(spec, tuning=t3)
    assert e.stats["launches_per_step"] == 3
    e.close()
    three = _run(spec, "fused3", 15, False, tuning=t3)
    _bits_equal(two, three, case)


@pytest.mark.parametrize("mode,ncond,cw,tc", [("vjoint", 1, "37", None), ("vjoint", 2, "29", None), ("phase", 1, "41", None),
                                              ("vjoint", 1, None, "1024"), ("phase", 1, None, "1024"), ("vjoint", 2, None, None)])
def test_two_launch_step_medium_sizes(mode, ncond, cw, tc):
    """The same at 3001 (x n_conditions) cells x 300 genes -- several gene blocks, many cell blocks, ragged tiles, two samples
    (6 angular-speed coefficients: rows of 8 floats), 1024-cell blocks -- over 25 steps, and through a hipGraph replay."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    # K_main's own nu_omega partials even where they cost the 4-genes-per-lane S+U kernel a resident workgroup (the engine
    # declines that trade above 12 cells per wave: profiles/r04_small_shard.md), and the dense histogram tables on both sides
    t2 = Tuning(cells_per_wave=int(cw) if cw else 0, tail_cells=int(tc) if tc else 0, pw_inline="force", hist_dense="dense")
    spec = make_phase_spec(3001, 300, seed=5) if mode == "phase" else make_velocity_spec(3001, 300, mode, n_conditions=ncond, Hw=1, seed=5)
    e = HipEngine(spec, tuning=t2)
    assert e.stats["launches_per_step"] == 2 and (mode == "phase" or e.stats["pw_inline"] == (4 if ncond == 1 else 8)), e.stats
    e.close()
    two = _run(spec, "fused3", 25, False, tuning=t2)
    two_g = _run(spec, "fused3", 25, True, tuning=t2)
    three = _run(spec, "fused3", 25, False, tuning=t2.replace(tail2=False))
    _bits_equal(two, three, f"{mode} x{ncond}")
    _bits_equal(two_g, three, f"{mode} x{ncond} (graph)")
    if mode != "phase":
        # ... and against the cell blocks' own partial sums of d loglik / d nu_omega (pw_inline="off": three launches, nothing
        # from K_main): only the association of one sum over the cells differs -- float32 rounding after 2 steps
        a = _run(spec, "fused3", 2, False, tuning=t2)
        b = _run(spec, "fused3", 2, False, tuning=t2.replace(pw_inline="off"))
        # ... and with the (value, multiplicity) lists instead of the dense tables
        c = _run(spec, "fused3", 2, False, tuning=t2.replace(pw_inline="off", hist_dense="lists"))
        assert np.allclose(c["l"], b["l"], rtol=5e-7, atol=0)
        _same(c["p"], b["p"], "params after 2 steps, lists vs dense tables", rtol=2e-5, atol=2e-6)
        assert np.allclose(a["l"], b["l"], rtol=2e-7, atol=0)
        _same(a["p"], b["p"], "params after 2 steps", rtol=2e-6, atol=2e-7)
        _same(a["g"][4:], b["g"][4:], "gradient of step 2", rtol=1e-5, atol=1e-5)


def test_two_launch_step_with_highly_expressed_genes():
    """Round 6 (`vc_stats.hist_split`): a gene block whose largest count exceeds 255 has its dense histogram sums evaluated by four
    QUARTER blocks of the one-launch tail (16 genes x 16 slices of the count axis, one wave per SIMD) -- the same per-slice sums, the
    slices of a gene added in the same order.  Genes in three gene blocks get counts beyond 255, one of them beyond the 640 levels the
    quarter blocks prefetch (a second trip over the table), one block keeps its small counts; over 25 steps the two-launch step must
    equal the three-launch step (K_tail / K_omega: one 4-wave block per gene block, `vc_hist_dense_block`) bit for bit, and the unfused
    sequence to the usual rounding."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3001, 300, "vjoint", n_conditions=1, Hw=1, seed=5)
    for mat, gene, factor in ((spec.S, 3, 60.0), (spec.U, 3, 25.0), (spec.S, 70, 25.0), (spec.U, 133, 400.0), (spec.S, 299, 30.0)):
        mat[gene] = torch.clamp(torch.floor(mat[gene] * factor), max=2000.0)           # integer counts below the tables' 2048 levels
    assert float(spec.S.max()) > 300 and float(spec.U.max()) > 700
    t2 = Tuning(cells_per_wave=37, pw_inline="force", hist_dense="dense")
    e = HipEngine(spec, tuning=t2)
    assert e.stats["launches_per_step"] == 2 and e.stats["hist_split"] >= 4, e.stats       # S: blocks 0, 1, 4; U: blocks 0, 2
    e.close()
    two = _run(spec, "fused3", 25, False, tuning=t2)
    three = _run(spec, "fused3", 25, False, tuning=t2.replace(tail2=False))
    _bits_equal(two, three, "highly expressed genes")
    assert two["status"][0] and np.isfinite(two["l"]).all()
    ref = _run(spec, "fused", 25, False, tuning=t2)
    assert np.allclose(two["l"], ref["l"], rtol=2e-6, atol=0)
    _same(two["p"], ref["p"], "params, two launches vs the unfused sequence", rtol=5e-5, atol=5e-6)


def test_two_launch_step_resumes_and_mixes_with_step_with_loss():
    """Checkpoint / resume into a new engine and run_perf mixed with step_with_loss on the two-launch step: the same
    trajectory bit for bit (the histogram halves, the shape_inv snapshot and the nu_omega snapshot are re-primed)."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3001, 300, "vjoint", n_conditions=1, Hw=1, seed=6)
    from velocycle_amd.tuning import Tuning
    _resume_body(spec, Tuning(pw_inline="force"))


def _resume_body(spec, tun):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    ref = _run(spec, "fused3", 13, False, tuning=tun)
    e1 = HipEngine(spec, tuning=tun)
    assert e1.stats["launches_per_step"] == 2
    r1 = SVIRunner(e1, OPT, mode="perf", seed=7)
    r1.run_perf(4)
    l = r1.perf_losses() + [r1.step_with_loss() for _ in range(3)]
    sd = r1.state_dict()
    e2 = HipEngine(spec, tuning=tun)
    r2 = SVIRunner(e2, OPT, mode="perf", seed=7)
    r2.load_state_dict(sd)
    r2.run_perf(6)
    assert torch.equal(e2.params.cpu(), ref["p"]) and np.array_equal(np.array(r2.perf_losses()), ref["l"])
    assert l == list(ref["l"][:7])
    e1.close(); e2.close()
