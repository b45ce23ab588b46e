"""GPU: the two BASELINE.json configurations that round 1 left unexercised at their stated size (SURVEY.md §8d stand-ins).

configs[4]  "Aissa PC9 two-sample (d0+d3) joint velocity_inference with per-condition angularspeed, 2 MI355X":
            synthetic 2 x 5 000 cells x 500 genes, Nx = Nb = 2, with_delta_nu -- one step and a 40-step trajectory against
            the float64 oracle, and the same problem split over rank / world_size = 2 engines (summed gradients equal
            the single engine's; the two-process run of it is tests/test_hip_fit_sharded.py).
configs[0]  "Capolupo one-sample phase_inference, ~3k cells": synthetic 3 000 cells x 200 genes phase_inference -- one step
            and a 40-step trajectory against the oracle.
Tolerances: tests/helpers.py (loss 1e-5 relative; gradients 2e-3 of the block max-norm or 4x the float32 oracle's own
error; trajectories within 4x the float32 oracle's drift, posterior means within 1e-3 of the block max-norm)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu

OPT = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / 1000), "betas": (0.80, 0.99)}


def _mk(spec, **kw):
    from velocycle_amd.engine import HipEngine
    return HipEngine(spec, **kw)


@pytest.fixture(scope="module", params=["vjoint", "vcond"])
def two_sample(request):
    from velocycle_amd.workloads import make_velocity_spec
    spec = make_velocity_spec(5000, 500, request.param, n_conditions=2, Hw=1, seed=6)
    assert spec.Nc == 10000 and spec.Nx == 2 and spec.Nb == 2 and spec.with_delta_nu
    if request.param == "vcond":
        assert set(spec.condition_on) == {"ϕxy", "ν", "shape_inv", "Δν"}
    return spec


def test_two_sample_step_matches_oracle(two_sample):
    from velocycle_amd.rng import draw_eps
    spec = two_sample
    eng = _mk(spec)
    g = torch.Generator().manual_seed(3)
    first = draw_eps(spec, g)
    eng.init_params(first.get("_cov_factor_draw"))
    eps = draw_eps(spec, g)
    eng.elbo_grad(eps=eng.pack_eps(eps))
    H.assert_step_matches_oracle(eng, spec, eps)
    eng.close()


def test_two_sample_trajectory_matches_oracle(two_sample):
    from velocycle_amd.svi import SVIRunner
    spec = two_sample
    eng = _mk(spec)
    run = SVIRunner(eng, OPT, mode="parity", seed=13)
    losses, snaps = [], {}
    for t in range(40):
        if t < 6 or t in (12, 25, 39):      # teacher forcing (helpers.py): the parameters each of these steps starts from
            snaps[t] = {k: v.clone() for k, v in eng.named().items()}
        losses.append(run.step())
    H.assert_trajectory_within_float32_spread(spec, OPT, 40, 13, losses, eng.named(), snapshots=snaps)
    eng.close()


def test_two_sample_split_over_two_ranks_equals_single_engine(two_sample):
    """rank / world_size = 2 engines on one GPU (cells 0..4999 = sample d0, 5000..9999 = sample d3): the sum of the
    two all-reduce buffers equals the single engine's, per-cell gradients equal its slices; then 10 SVI steps with the
    summed buffer fed to both shards' optimisers track the single engine (what the RCCL all-reduce does per step)."""
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.svi import FlatClippedAdam
    spec = two_sample
    g = torch.Generator().manual_seed(4)
    first = draw_eps(spec, g)
    cov = first.get("_cov_factor_draw")
    full = _mk(spec)
    shards = [_mk(spec, rank=r, world_size=2) for r in range(2)]
    for e in [full] + shards:
        e.init_params(cov)
    nrep = full.header + full.n_global
    opts = [FlatClippedAdam(e.total - e.header, OPT, e.device) for e in [full] + shards]
    for step in range(10):
        eps = draw_eps(spec, g)
        for e in [full] + shards:
            e.elbo_grad(eps=e.pack_eps(eps))
        torch.cuda.synchronize()
        tot = sum(s.grad[:nrep].double() for s in shards)
        ref = full.grad[:nrep].double()
        lsum, lref = float(tot[0] + tot[1]), float(ref[0] + ref[1])
        assert abs(lsum - lref) <= 1e-6 * abs(lref), (step, lsum, lref)
        assert abs(lref - full.loss()) <= 1e-6 * abs(lref)
        # step 0: identical parameters on both sides -> only the reassociation of the two partial sums differs; later
        # steps: the optimiser has amplified that rounding a little (Adam's m / sqrt(v) where a gradient is near zero)
        # (round 4: the single engine takes the partials of d loglik / d nu_omega from K_main -- another association of that
        # one sum than the shards' cell blocks -- and a gene on the relu kink of ElogU turns such a 1e-7 into per cent of ITS
        # gradient within ten steps: 9e-8 / 9e-8 / 4e-6 / 9e-6 / 4e-6 / 6e-5 / 1e-3 / 1e-4 / 2e-5 / 6e-3 of the largest element
        # were measured step by step; the sharded STEP itself is held against the single-rank step in test_hip_sharded_step.py)
        # The bar is 5e-3 of the largest element at every step after the first (it had been widened to 2e-2 from
        # step 6 on).  What exceeds it late in the run is NOT the gradient evaluation but single genes on the relu kink of ElogU
        # whose parameters have separated between the two float32 trajectories: those elements are named, must be fewer than
        # 0.1 % of the buffer, and must stay below 2e-2 -- everything else holds 5e-3 (1e-3 up to step 5, 2e-4 at step 0).
        tol = 2e-4 if step == 0 else (1e-3 if step <= 5 else 5e-3)
        scale = max(float(ref[4:].abs().max()), 1e-3)
        dev = (tot[4:] - ref[4:]).abs() / scale
        over = (dev > tol).nonzero().reshape(-1)
        if over.numel():
            assert step > 5, (step, float(dev.max()))
            where = []
            for i in over.tolist()[:8]:
                off = i + 4
                name = next((n for n, (o, sz) in full.param_slices.items() if o <= off < o + sz), "?")
                where.append((name, off - full.param_slices[name][0] if name != "?" else off, float(dev[i])))
            print(f"[two-sample split] step {step}: {over.numel()} of {dev.numel()} gradient elements beyond {tol:g} of the largest "
                  f"(relu-kink genes of two separated float32 trajectories): {where}")
            assert over.numel() <= max(1, dev.numel() // 1000) and float(dev.max()) <= 2e-2, (step, over.numel(), float(dev.max()))
        xy = torch.cat([s.view(s.grad, "ϕxy_locs") for s in shards])
        xyf = full.view(full.grad, "ϕxy_locs")
        dxy = (xy - xyf).abs() / max(float(xyf.abs().max()), 1.0)
        if float(dxy.max()) > tol:           # the same bar, the same loud exception: named cells, < 0.1 % of them, below 2e-2
            bad = (dxy > tol).nonzero()
            print(f"[two-sample split] step {step}: {bad.shape[0]} of {dxy.numel()} phi_xy gradient elements beyond {tol:g}: "
                  f"{[(int(i), int(j), float(dxy[i, j])) for i, j in bad.tolist()[:8]]}")
            assert step > 5 and bad.shape[0] <= max(1, dxy.numel() // 1000) and float(dxy.max()) <= 2e-2, (step, bad.shape[0], float(dxy.max()))
        for s in shards:                    # the all-reduce: every rank continues from the summed buffer
            s.grad[:nrep] = tot.float()
        for e, o in zip([full] + shards, opts):
            o.step(e.params[e.header:], e.grad[e.header:])
    glob = full.params[full.header:nrep]
    for s in shards:
        a, b = s.params[s.header:nrep], glob
        fin = torch.isfinite(b)
        # two float32 trajectories of ten steps (see above): in Adam's first steps the update of an element is ~ lr x sign(g), so an
        # element whose tiny gradient changes sign between the two runs moves apart by 2 lr per step -- the runs agree where
        # that does not happen: 99 % of the elements to 1e-3, the typical distance far below
        diff = (a[fin] - b[fin]).abs()
        assert float((diff <= 1e-3 + 1e-3 * b[fin].abs()).float().mean()) >= 0.99, float((diff <= 1e-3 + 1e-3 * b[fin].abs()).float().mean())
        assert float(diff.median()) <= 1e-5
    for e in [full] + shards:
        e.close()


# ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def phase_3k():
    from velocycle_amd.workloads import make_phase_spec
    return make_phase_spec(3000, 200, seed=5)


def test_phase_3k_step_matches_oracle(phase_3k):
    from velocycle_amd.rng import draw_eps
    from velocycle_amd.tuning import Tuning
    spec = phase_3k
    for cw in (0, 29):                        # the balanced one-round tiling and a ragged many-chunk one
        eng = _mk(spec, tuning=Tuning(cells_per_wave=cw))
        assert "phase" in eng.stats["main_kernel"]
        g = torch.Generator().manual_seed(8)
        draw_eps(spec, g)
        eng.init_params()
        eps = draw_eps(spec, g)
        eng.elbo_grad(eps=eng.pack_eps(eps))
        H.assert_step_matches_oracle(eng, spec, eps)
        eng.close()


def test_phase_3k_trajectory_matches_oracle(phase_3k):
    from velocycle_amd.svi import SVIRunner
    spec = phase_3k
    eng = _mk(spec)
    run = SVIRunner(eng, OPT, mode="parity", seed=17)
    losses, snaps = [], {}
    for t in range(40):
        if t < 6 or t in (12, 25, 39):
            snaps[t] = {k: v.clone() for k, v in eng.named().items()}
        losses.append(run.step())
    H.assert_trajectory_within_float32_spread(spec, OPT, 40, 17, losses, eng.named(), snapshots=snaps)
    eng.close()


def test_phase_full_size_shard_additivity():
    """phase_inference at BASELINE's bench size (50k x 2k): no oracle at this size, so the size-independent properties --
    bitwise repeatability and additivity of 4 cell shards."""
    from velocycle_amd.workloads import make_phase_spec
    spec = make_phase_spec(50000, 2000, seed=0, device="cuda")
    full = _mk(spec)
    full.init_params()
    full.elbo_grad(eps=None, seed=5, step=2)
    torch.cuda.synchronize()
    g0, loss = full.grad.clone(), full.loss()
    full.elbo_grad(eps=None, seed=5, step=2)
    torch.cuda.synchronize()
    assert torch.equal(g0, full.grad) and full.loss() == loss
    nrep = full.header + full.n_global
    ref = full.grad[:nrep].double().cpu()
    tot = torch.zeros_like(ref)
    xy = []
    for r in range(4):
        s = _mk(spec, rank=r, world_size=4)
        s.init_params()
        s.elbo_grad(eps=None, seed=5, step=2)
        torch.cuda.synchronize()
        tot += s.grad[:nrep].double().cpu()
        xy.append(s.view(s.grad, "ϕxy_locs").cpu())
        s.close()
    assert abs(float(tot[0] + tot[1]) - loss) <= 2e-7 * abs(loss)
    assert float((tot[4:] - ref[4:]).abs().max()) <= 2e-5 * float(ref[4:].abs().max())
    assert torch.allclose(torch.cat(xy), full.view(full.grad, "ϕxy_locs").cpu(), rtol=1e-4, atol=1e-3)
    full.close()
