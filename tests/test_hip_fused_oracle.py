"""GPU: the BENCHMARKED path held directly against the oracle (VERDICT r2 "missing" #2).

bench.py times `SVIRunner(mode="perf")` = vc_svi_run_fused (K_main -> K_tail -> K_omega per step, eps from the in-kernel
Philox stream).  Until round 3 that path was only compared with the unfused HIP sequence.  Here, for every step t of a
fused run, the eps vector of (seed, t) is rebuilt by an independent engine's sampling kernel, and the float64 oracle
(oracle/velocycle_oracle.py: loss_and_grads + ClippedAdam, the restatement of velocity_inference_model.py:118-121 /
phase_inference_model.py:166-170 with pyro's Trace_ELBO and ClippedAdam) is replayed from the same initial parameters on
exactly those draws.  Bars: loss[t] within 1e-5 relative for the first 5 steps; afterwards float32 and float64 Adam
trajectories separate by themselves, so the yardstick is the oracle's own float32 replay (x4); fitted parameters within 1e-3
of each block's max-norm wherever the float32 oracle is (tests/helpers.py: assert_params_track_oracle -- single genes that
cross the relu kink of ElogU may leave the float64 trajectory in ANY float32 run)."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
OPT = {"lr": 0.03, "lrd": (0.005 / 0.03) ** (1.0 / 1000), "betas": (0.80, 0.99)}


def _fused_vs_oracle(spec, n, seed, check_params=True, report=None, tuning=None):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    eng = HipEngine(spec, tuning=tuning)
    run = SVIRunner(eng, OPT, mode="perf", seed=seed)
    assert run.adam_impl == "fused3" and not run.use_graph          # the path bench.py times
    flat0 = eng.params.detach().clone()
    par0 = {k: v.detach().cpu().clone() for k, v in eng.named().items()}
    run.run_perf(n)
    losses = np.array(run.perf_losses())
    assert eng.status() == (True, -1, 0) and len(losses) == n
    got = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in eng.named().items()}
    eng.close()
    eps = H.philox_eps_list(spec, flat0, seed, n)
    l64, par64 = H.oracle_replay(spec, OPT, par0, eps, torch.float64)
    l32, par32 = H.oracle_replay(spec, OPT, par0, eps, torch.float32)
    l64, l32 = np.array(l64), np.array(l32)
    rel_hip, rel_32 = np.abs(losses - l64) / np.abs(l64), np.abs(l32 - l64) / np.abs(l64)
    assert rel_hip[:5].max() <= 1e-5, rel_hip[:5]
    assert (rel_hip <= np.maximum(1e-5, 4 * np.maximum.accumulate(rel_32))).all(), (rel_hip.max(), rel_32.max())
    if check_params:
        H.assert_params_track_oracle(got, {k: v.numpy() for k, v in par64.items()}, {k: v.double().numpy() for k, v in par32.items()},
                                     report=report)
    return rel_hip


@pytest.mark.parametrize("case", ["vel_mf_joint", "vel_lrmn_cond", "phase_nb", "vel_mf_joint_dnu2", "vel_lrmn_joint"])
def test_fused_philox_run_matches_oracle_replay_on_fixtures(case):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    _fused_vs_oracle(H.spec_from_fixture(z), n=25, seed=1234)


@pytest.mark.parametrize("mode,ncond,gpl", [("vjoint", 1, 8), ("vjoint", 1, 4), ("vcond", 2, 8), ("vjoint", 2, 4)])
def test_fused_philox_run_matches_oracle_replay_medium(mode, ncond, gpl):
    """3001 (x conditions) cells x 300 genes: two gene blocks, ragged cell tiles, Nx = Nb = 2 with per-batch offsets; the
    8-genes-per-lane kernels (what the full-size benchmark runs; shards this small default to 4) and the 4-genes-per-lane ones."""
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    _fused_vs_oracle(make_velocity_spec(3001, 300, mode, n_conditions=ncond, Hw=1, seed=5), n=12, seed=77, report=f"{mode} x{ncond}",
                     tuning=Tuning(genes_per_lane=gpl))


def test_fused_philox_run_matches_oracle_replay_phase_medium():
    from velocycle_amd.workloads import make_phase_spec
    _fused_vs_oracle(make_phase_spec(3000, 200, seed=5), n=12, seed=3)


@pytest.mark.parametrize("mode", ["vjoint", "vcond"])
def test_fused_philox_run_matches_oracle_on_a_slice_of_the_benchmark_data(mode):
    """The first 2 000 cells of the 50 000 x 2 000 benchmark workload (bench.py's generator and seed): every gene block of
    the full-size launch, the oracle finishes in seconds."""
    import copy
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.workloads import make_velocity_spec
    gpl8 = Tuning(genes_per_lane=8)              # the kernel of the full-size launch (a 2 000-cell shard alone would take 4)
    full = make_velocity_spec(50000, 2000, mode, 1, 1, seed=0, device="cuda")
    n = 2000
    spec = copy.copy(full)
    spec.S, spec.U = full.S[:, :n].contiguous(), full.U[:, :n].contiguous()
    spec.count_factor, spec.Db, spec.D = full.count_factor[:n], full.Db[:, :n], full.D[:, :n]
    spec.phixy_prior = full.phixy_prior[:n]
    spec.condition_on = {k: (v[:n] if k == "ϕxy" else v) for k, v in full.condition_on.items()}
    del full
    _fused_vs_oracle(spec, n=6, seed=0, tuning=gpl8)


def test_sharded_sequence_matches_oracle_on_a_slice_of_the_benchmark_data():
    """The sharded sequence (K_main -> phase A -> exchange -> phase B, world = 2, in-process ranks) on the first 2 000 cells of
    the 50 000 x 2 000 V-joint benchmark data, replayed step by step by the float64 oracle on the Philox draws it used
    (the path SCALE runs had only been held against the single-rank step, never against the oracle)."""
    import copy
    from velocycle_amd import _lib
    from velocycle_amd.workloads import make_velocity_spec
    from tests.test_hip_sharded_step import _Rank, OPT as SOPT
    full = make_velocity_spec(50000, 2000, "vjoint", 1, 1, seed=0, device="cuda")
    nc = 2000
    spec = copy.copy(full)
    spec.S, spec.U = full.S[:, :nc].contiguous(), full.U[:, :nc].contiguous()
    spec.count_factor, spec.Db, spec.D = full.count_factor[:nc], full.Db[:, :nc], full.D[:, :nc]
    spec.phixy_prior = full.phixy_prior[:nc]
    del full
    world, n, seed = 2, 6, 0
    ranks = [_Rank(spec, r, world, seed) for r in range(world)]
    par0 = {k: v.detach().cpu().clone() for k, v in ranks[0].e.named().items()}
    par0["ϕxy_locs"] = torch.cat([r.e.view(r.e.params, "ϕxy_locs").detach().cpu() for r in ranks])
    for t in range(n):
        for r in ranks:
            r.call(seed, _lib.VC_PHASE_A, prime=(t == 0))
        torch.cuda.synchronize()
        tot = ranks[0].x + ranks[1].x
        for r in ranks:
            r.x.copy_(tot)
            r.call(seed, _lib.VC_PHASE_B)
    torch.cuda.synchronize()
    losses = ranks[0].ring[:n].cpu().numpy()
    got = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in ranks[0].e.named().items()}
    got["ϕxy_locs"] = torch.cat([r.e.view(r.e.params, "ϕxy_locs").detach().cpu() for r in ranks]).numpy().astype(np.float64)
    for r in ranks:
        assert r.e.status() == (True, -1, 0)
        r.e.close()
    # the single-engine layout of the same parameters for the independent eps rebuild
    from velocycle_amd.engine import HipEngine
    e0 = HipEngine(spec)
    e0.set_params(par0)
    flat0 = e0.params.detach().clone()
    e0.close()
    opt = {"lr": SOPT["lr"], "lrd": SOPT["lrd"], "betas": (SOPT["b1"], SOPT["b2"])}
    eps = H.philox_eps_list(spec, flat0, seed, n)
    l64, par64 = H.oracle_replay(spec, opt, par0, eps, torch.float64)
    l32, par32 = H.oracle_replay(spec, opt, par0, eps, torch.float32)
    l64, l32 = np.array(l64), np.array(l32)
    rel_hip, rel_32 = np.abs(losses - l64) / np.abs(l64), np.abs(l32 - l64) / np.abs(l64)
    assert rel_hip[:5].max() <= 1e-5, rel_hip[:5]
    assert (rel_hip <= np.maximum(1e-5, 4 * np.maximum.accumulate(rel_32))).all(), (rel_hip.max(), rel_32.max())
    H.assert_params_track_oracle(got, {k: v.numpy() for k, v in par64.items()}, {k: v.double().numpy() for k, v in par32.items()},
                                 report="sharded world=2, 2000-cell slice")
