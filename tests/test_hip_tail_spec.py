"""GPU: the small kernels compiled for a configuration's SIGNATURE (csrc/vc_tail_spec.h, round 5) against the run-time-flag kernels
they specialise.  The specialised instantiation is the same source with the signature's fields as compile-time facts -- dead
branches removed, loops unrolled, every remaining statement in the same order -- so parameters, optimiser moments, gradients and
losses must be EQUAL BIT FOR BIT, on every row of csrc/vc_tail_spec_rows.inc, under each launch structure the row is compiled for
(one-launch tail, merged tail of the tutorial flow, phases A / B of a sharded rank).  Also: the rows are keyed on the whole
signature -- a configuration that differs in one field runs the run-time-flag kernels."""
import numpy as np
import pytest
import torch

from tests.test_hip_fused import _bits_equal, _run
from tests.test_hip_sharded_step import _run_sharded

pytestmark = pytest.mark.gpu

ROWS = {            # name of the row a workload must select: (builder, mode, samples[, omega harmonics Hw = 1])
    "vjoint": ("vel", "vjoint", 1), "vcond": ("vel", "vcond", 1), "phase": ("phase", None, 1),
    "vjoint_lrmn": ("vel", "vjoint_lrmn", 1), "vcond_mf": ("vel", "vcond_mf", 1),
    # round 6: the tutorials' constant-omega first velocity stage (omega_n_harmonics = 0), one sample / two / any number
    "vcond_hw0": ("vel", "vcond", 1, 0), "vcond_hw0_2s": ("vel", "vcond", 2, 0), "vcond_hw0_multi": ("vel", "vcond", 3, 0),
    # round 6: exactly two samples (BASELINE configs[4]) with every count closed ...
    "vjoint_2s": ("vel", "vjoint", 2), "vcond_2s": ("vel", "vcond", 2),
    # ... in front of the "multi" rows that leave the number of batches / conditions open (three samples here; two where no closed row exists)
    # (three conditions x three omega coefficients are more than K_main's own nu_omega partials carry -- three launches, no single-rank
    # row: the open rows are reached with three BATCHES under two conditions)
    "vjoint_multi": ("vel", "vjoint", 2, 1, 3), "vcond_multi": ("vel", "vcond", 2, 1, 3), "phase_multi": ("phase", None, 3),
    "vjoint_lrmn_multi": ("vel", "vjoint_lrmn", 2), "vcond_mf_multi": ("vel", "vcond_mf", 2),
    # round 6: two harmonics of the expression map -- the default n_harmonics of preprocess_for_* (preprocessing.py:108,217); V-joint with
    # five coefficients per gene runs three launches on one rank (no single-rank row), its rank row exists
    "vcond_h2": ("vel", "vcond", 1, 1, None, 2), "phase_h2": ("phase", None, 1, 1, None, 2),
}
RANK_ROWS = {"vjoint_rank": ("vel", "vjoint", 1), "vcond_rank": ("vel", "vcond", 1), "phase": ("phase", None, 1),
             "vjoint_lrmn_rank": ("vel", "vjoint_lrmn", 1), "vcond_mf_rank": ("vel", "vcond_mf", 1),
             "vcond_hw0_rank": ("vel", "vcond", 1, 0), "vcond_hw0_2s_rank": ("vel", "vcond", 2, 0), "vcond_hw0_multi_rank": ("vel", "vcond", 3, 0),
             "vjoint_2s_rank": ("vel", "vjoint", 2), "vcond_2s_rank": ("vel", "vcond", 2),
             "vjoint_multi_rank": ("vel", "vjoint", 3), "vcond_multi_rank": ("vel", "vcond", 3), "phase_multi": ("phase", None, 2),
             "vjoint_lrmn_multi_rank": ("vel", "vjoint_lrmn", 2), "vcond_mf_multi_rank": ("vel", "vcond_mf", 2),
             "vjoint_h2_rank": ("vel", "vjoint", 1, 1, None, 2), "vcond_h2_rank": ("vel", "vcond", 1, 1, None, 2), "phase_h2": ("phase", None, 1, 1, None, 2)}


def _spec(kind, mode, ncond, hw=1, nbatch=None, H=1, nc=2100, ng=260):
    from velocycle_amd.workloads import make_phase_spec, make_velocity_spec
    if kind == "phase":
        return make_phase_spec(nc // ncond, ng, seed=3, n_batches=ncond, H=H)
    return make_velocity_spec(nc // (nbatch or ncond), ng, mode, n_conditions=ncond, Hw=hw, seed=3, n_batches=nbatch, H=H)


@pytest.mark.parametrize("row", sorted(ROWS))
def test_single_rank_specialisation_is_bit_identical(row):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.tuning import Tuning
    spec = _spec(*ROWS[row])
    e = HipEngine(spec)
    assert e.stats["tail_spec_name"] == row and e.stats["tail_spec"] > 0 and e.stats["launches_per_step"] == 2, e.stats
    e.close()
    e = HipEngine(spec, tuning=Tuning(no_tail_spec=True))
    assert e.stats["tail_spec_name"] == "generic" and e.stats["tail_spec"] == 0
    e.close()
    for use_graph in (False, True):
        a = _run(spec, "fused3", 12, use_graph)
        b = _run(spec, "fused3", 12, use_graph, tuning=Tuning(no_tail_spec=True))
        _bits_equal(a, b, row)


@pytest.mark.parametrize("row", sorted(RANK_ROWS))
def test_sharded_rank_specialisation_is_bit_identical(row):
    from velocycle_amd.tuning import Tuning
    spec = _spec(*RANK_ROWS[row])
    n, seed, world = 6, 5, 3
    a = _run_sharded(spec, world, n, seed)
    b = _run_sharded(spec, world, n, seed, tuning=Tuning(no_tail_spec=True))
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    for ra, rb in zip(a, b):
        assert ra.e.stats["tail_spec_name"] == row and rb.e.stats["tail_spec_name"] == "generic", (ra.e.stats, rb.e.stats)
        assert torch.equal(nz(ra.e.params), nz(rb.e.params)) and torch.equal(ra.m, rb.m) and torch.equal(ra.v, rb.v)
        assert torch.equal(ra.ring[:n], rb.ring[:n]) and int(ra.sd.item()) == int(rb.sd.item()) == n
        assert ra.e.status() == rb.e.status() == (True, -1, 0)
    for r in a + b:
        r.e.close()


@pytest.mark.parametrize("row", ["vjoint", "vcond", "phase", "vjoint_multi", "vjoint_lrmn", "vcond_hw0", "vjoint_2s"])
def test_particle_step_specialisation_is_bit_identical(row):
    """vc_svi_run_particles (K_pre / K_post of all particles, K_fin + average + optimiser: K + 3 launches) in the instantiations compiled
    for the row against the run-time-flag kernels: K = 3, eight steps, parameters / moments / losses bit for bit."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.tuning import Tuning
    spec = _spec(*ROWS[row])
    out = []
    for tun in (None, Tuning(no_tail_spec=True)):
        e = HipEngine(spec, tuning=tun)
        assert e.stats["tail_spec_name"] == (row if tun is None else "generic")
        r = SVIRunner(e, {"lr": 0.03, "lrd": 0.995, "betas": (0.8, 0.99)}, mode="perf", seed=9, num_particles=3)
        r.run_perf(8)
        out.append((e.params.clone().cpu(), r.opt.m.clone().cpu(), r.opt.v.clone().cpu(), np.array(r.perf_losses()), e.status()))
        e.close()
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    a, b = out
    assert torch.equal(nz(a[0]), nz(b[0])) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert a[4] == b[4] == (True, -1, 0) and len(a[3]) == 8


@pytest.mark.parametrize("case", ["vel_mf_poisson", "vel_mf_lognormal", "phase_poisson"])
def test_a_signature_without_a_row_runs_the_generic_kernels(case):
    """A Poisson or Lognormal noise model: configurations the library has no compiled row for keep the run-time-flag kernels (and say
    so); the signature is 27 ints."""
    import os
    from tests import helpers as H
    from velocycle_amd.engine import HipEngine
    path = f"{H.GOLDEN}/ref_step_{case}.npz"
    if not os.path.exists(path):
        pytest.skip(f"no fixture {case}")
    spec = H.spec_from_fixture(H.load_fixture(path))
    e = HipEngine(spec)
    assert e.stats["tail_spec_name"] == "generic" and e.stats["tail_spec"] == 0, e.stats
    assert len(e.signature()) == 27
    e.close()
