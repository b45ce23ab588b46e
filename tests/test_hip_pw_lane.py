"""GPU: the tutorial flow's likelihood kernel with the nu_omega partials kept PER LANE (round 6, `vc_stats.pw_lane`).

One condition with D == 1 (every one-sample fit): the W row of a cell, D[x,c] zeta_omega_h(phi_c), is (1, sin k phi_c, cos k phi_c) --
what the cell's record already holds.  The U-only kernel then accumulates A3_c W_c per lane (1 + 2 Hw plain VALU per cell) instead of
reducing A3 over the wave per cell (64-lane DPP tree + staging), and stores no per-cell rows.  Same sums in another order: held
against the float64 oracle (one step, every gradient block) and against the wave-reduction path (`Tuning(pw_lane=False)`) over a
run, for omega with one harmonic and with none (the tutorials' first velocity stage)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
OPT = {"lr": 0.03, "lrd": 0.995, "betas": (0.8, 0.99)}


@pytest.mark.parametrize("mode,hw", [("vcond", 1), ("vcond", 0), ("vcond_mf", 1)])
def test_per_lane_partials_match_oracle_and_the_wave_reduction(mode, hw):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(3001, 300, mode, n_conditions=1, Hw=hw, seed=4)
    e = HipEngine(spec)
    assert e.stats["pw_lane"] and e.stats["pw_inline"] == 4 and e.stats["launches_per_step"] == 2, e.stats
    g = torch.Generator().manual_seed(2)
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


def test_two_conditions_or_a_design_that_is_not_all_ones_keep_the_wave_reduction():
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.workloads import make_velocity_spec
    e = HipEngine(make_velocity_spec(1500, 200, "vcond", n_conditions=2, Hw=1, seed=4))
    assert not e.stats["pw_lane"] and e.stats["pw_inline"] == 8
    e.close()
    spec = make_velocity_spec(3000, 200, "vcond", n_conditions=1, Hw=1, seed=4)
    spec.D = spec.D.clone()
    spec.D[0, 5] = 0.5
    e = HipEngine(spec)
    assert not e.stats["pw_lane"] and e.stats["pw_inline"] == 4
    e.close()
