"""GPU: the likelihood kernel's cell tiling with unequal shares per dispatch pass (vc_engine.hip).

The shares only move the boundaries between the waves' cell ranges: every cell still belongs to exactly one wave of every
gene block.  At a size where the grid spans all passes and the shares are in force (30 000 cells x 128 genes: 512 / 768
workgroups, >= 12 cells per wave) one ELBO + gradient evaluation is compared with the float64 oracle, and the same
evaluation under other shares (equal, strongly skewed, three and four entries) with the default: per-cell gradients to
float32 rounding of the per-cell sums, gene-level gradients and the loss to the rounding of a re-associated sum."""
import os

import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _evaluate(spec, eps_seed, shares, **tun):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.tuning import Tuning
    eng = HipEngine(spec, tuning=Tuning(pass_shares=shares, **tun))          # the tiling is fixed at vc_finalize
    g = torch.Generator().manual_seed(eps_seed)
    first = draw_eps(spec, g)
    eng.init_params(first.get("_cov_factor_draw"))
    eps = draw_eps(spec, g)
    eng.elbo_grad(eps=eng.pack_eps(eps))
    torch.cuda.synchronize()
    return eng, eps


@pytest.mark.parametrize("mode", ["vjoint", "vcond", "phase"])
def test_pass_shares_leave_the_result_alone(mode):
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    gpl8 = dict(genes_per_lane=8)              # the multi-pass kernels of the full-size problems (a shard this narrow defaults to 4)
    if mode == "vcond":
        # (round 6: with the nu_omega partials per lane the U-only kernel holds FOUR workgroups per CU -- 10 cells per wave here, below
        # the 12 from which the passes take unequal shares; the three-pass kernel this test was written on is the row-storing one)
        gpl8["pw_lane"] = False
    if mode == "phase":
        spec = make_phase_spec(40000, 128, seed=21)
    else:
        spec = make_velocity_spec(40000, 128, mode, 1, 1, seed=21)
    ref, eps = _evaluate(spec, 5, None, **gpl8)
    st = ref.stats
    assert st["main_grid"] > 256, st                      # more than one dispatch pass
    pc = st["pass_cells"]
    # defaults (round 3): the two-pass S+U kernel 0.75 : 0.25, the three-pass one-matrix kernels 0.62 : 0.26 : 0.12
    r = 1.0 / 3.0 if mode == "vjoint" else 0.42
    assert pc[0] > pc[1] and abs(pc[1] - r * pc[0]) <= 2, pc
    H.assert_step_matches_oracle(ref, spec, eps)
    g_ref = {k: v.double().cpu().clone() for k, v in ref.named(ref.grad).items()}
    loss_ref = float(ref.loss())
    for shares in ((1.0, 1.0), (0.85, 0.15), (0.5, 0.3, 0.2), (4.0, 3.0, 2.0, 1.0)):
        eng, _ = _evaluate(spec, 5, shares, **gpl8)
        pcs = eng.stats["pass_cells"]
        assert (pcs[0] == pcs[1]) == (shares == (1.0, 1.0)) and pcs != pc, (shares, pcs)
        assert abs(float(eng.loss()) - loss_ref) <= 2e-6 * abs(loss_ref), (shares, float(eng.loss()), loss_ref)
        for name, got in eng.named(eng.grad).items():
            want = g_ref[name]
            fin = torch.isfinite(want)
            err = (got.double().cpu()[fin] - want[fin]).abs().max()
            assert err <= 2e-5 * max(float(want[fin].abs().max()), 1e-3), (shares, name, float(err))
        eng.close()
    ref.close()


def test_pass_shares_with_gene_blocks_that_do_not_divide_the_cus():
    """1 100 genes = 3 gene blocks: the boundary between two dispatch passes then falls inside a row of chunks (chunk c of
    gene block gb runs in pass (3 c + gb) / CUs), so the gene blocks have different numbers of chunks per pass.  One
    evaluation against the float64 oracle, and against the balanced tiling selected the old way (Tuning.cells_per_wave), which
    does not go through the share arithmetic at all."""
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(9000, 1100, "vjoint", 1, 1, seed=23)
    ref, eps = _evaluate(spec, 6, None)
    pc = ref.stats["pass_cells"]
    assert ref.stats["main_grid"] > 256 and pc[0] > pc[1] >= 8, (ref.stats["main_grid"], pc)
    H.assert_step_matches_oracle(ref, spec, eps)
    g_ref = {k: v.double().cpu().clone() for k, v in ref.named(ref.grad).items()}
    loss_ref = float(ref.loss())
    old, _ = _evaluate(spec, 6, None, cells_per_wave=14)
    assert len(set(old.stats["pass_cells"])) == 1
    assert abs(float(old.loss()) - loss_ref) <= 2e-6 * abs(loss_ref)
    for name, got in old.named(old.grad).items():
        want = g_ref[name]
        fin = torch.isfinite(want)
        err = (got.double().cpu()[fin] - want[fin]).abs().max()
        assert err <= 2e-5 * max(float(want[fin].abs().max()), 1e-3), (name, float(err))
    old.close(); ref.close()
