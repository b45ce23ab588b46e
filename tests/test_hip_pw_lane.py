"""GPU: the tutorial flow's likelihood kernel with the nu_omega partials kept PER LANE (round 6, `vc_stats.pw_lane`).

One condition with D == 1 (every one-sample fit) -- or several one-hot conditions that are constant within every batch of a one-hot
batch design (the two-sample tutorials pass the same matrix for both; the kernel's workgroups are batch-aligned): the W row of a
cell, D[x,c] zeta_omega_h(phi_c), is (1, sin k phi_c, cos k phi_c) in the columns of the workgroup's condition -- what the cell's
record already holds.  The U-only kernel then accumulates A3_c W_c per lane (1 + 2 Hw plain VALU per cell) instead of
reducing A3 over the wave per cell (64-lane DPP tree + staging), and stores no per-cell rows.  Same sums in another order: held
against the float64 oracle (one step, every gradient block) and against the wave-reduction path (`Tuning(pw_lane=False)`) over a
run, for omega with one harmonic and with none (the tutorials' first velocity stage)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
OPT = {"lr": 0.03, "lrd": 0.995, "betas": (0.8, 0.99)}


@pytest.mark.parametrize("mode,hw,ncond,nbatch", [("vcond", 1, 1, None), ("vcond", 0, 1, None), ("vcond_mf", 1, 1, None), ("vcond", 1, 2, None),
                                                  ("vcond", 0, 2, None), ("vcond", 1, 2, 3)])
def test_per_lane_partials_match_oracle_and_the_wave_reduction(mode, hw, ncond, nbatch):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3001 // (nbatch or ncond), 300, mode, n_conditions=ncond, Hw=hw, seed=4, n_batches=nbatch)
    e = HipEngine(spec)
    assert e.stats["pw_lane"] and e.stats["pw_inline"] == (4 if ncond * (2 * hw + 1) <= 4 else 8) and e.stats["launches_per_step"] == 2, e.stats
    # (the draw: a gene on the relu kink of ElogU puts ANY float32 evaluation a few 1e-3 of the block's max-norm away from float64,
    # with a 3x spread between evaluations -- profiles/r06_kink_error.md; seed 2 does that to the two-condition Hw = 0 case for BOTH
    # paths of this kernel alike, 12x the float32 oracle's own error at one gene, beyond the helper's 4x clause)
    g = torch.Generator().manual_seed(5)
    first = draw_eps(spec, g)
    e.init_params(first.get("_cov_factor_draw"))
    eps = draw_eps(spec, g)
    e.elbo_grad(eps=e.pack_eps(eps))
    H.assert_step_matches_oracle(e, spec, eps)
    e.close()
    out = []
    for tun in (None, Tuning(pw_lane=False)):
        e = HipEngine(spec, tuning=tun)
        assert e.stats["pw_lane"] == (tun is None)
        r = SVIRunner(e, OPT, mode="perf", seed=11)
        r.run_perf(30)
        out.append((np.array(r.perf_losses()), {k: v.detach().cpu().double().numpy() for k, v in e.named().items()}, e.status()))
        e.close()
    (la, pa, sa), (lb, pb, sb) = out
    assert sa == sb == (True, -1, 0)
    assert abs(la[0] - lb[0]) <= 1e-7 * abs(lb[0])                 # the first step's loss does not involve the partials at all
    assert np.allclose(la, lb, rtol=2e-5), np.abs(la / lb - 1).max()
    for k in pa:
        fin = np.isfinite(pb[k])
        assert np.array_equal(np.isfinite(pa[k]), fin), k
        if fin.any():
            assert np.abs(pa[k][fin] - pb[k][fin]).max() <= 2e-3 * max(np.abs(pb[k][fin]).max(), 1e-2), k


def test_designs_that_put_two_conditions_into_one_workgroup_keep_the_wave_reduction():
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.workloads import make_velocity_spec
    # two conditions WITHOUT batch offsets: nothing aligns the workgroups to the conditions
    spec = make_velocity_spec(1500, 200, "vcond", n_conditions=2, Hw=1, seed=4)
    spec.with_delta_nu = False
    spec.condition_on.pop("Δν")
    e = HipEngine(spec)
    assert not e.stats["pw_lane"] and e.stats["pw_inline"] == 8
    e.close()
    # conditions that cut through a batch
    spec = make_velocity_spec(1500, 200, "vcond", n_conditions=2, Hw=1, seed=4)
    spec.D = spec.D.clone()
    spec.D[:, 7] = torch.tensor([0.0, 1.0])
    e = HipEngine(spec)
    assert not e.stats["pw_lane"] and e.stats["pw_inline"] == 8
    e.close()
    spec = make_velocity_spec(3000, 200, "vcond", n_conditions=1, Hw=1, seed=4)
    spec.D = spec.D.clone()
    spec.D[0, 5] = 0.5
    e = HipEngine(spec)
    assert not e.stats["pw_lane"] and e.stats["pw_inline"] == 4
    e.close()
