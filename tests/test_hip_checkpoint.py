"""GPU: checkpoint / resume of a fit (reference analogue pyro.get_param_store().get_state()/set_state(),
tutorials/1D_Pancreas_Analysis.ipynb cell 26): 20 steps == 10 steps + save + load into a NEW engine + 10 steps, bit for
bit, in the hipGraph performance path and in the host-eps parity path."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu
OPT = {"lr": 0.03, "lrd": 0.995, "betas": (0.8, 0.99)}


def _runner(spec, mode, **kw):
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.svi import SVIRunner
    e = HipEngine(spec)
    return e, SVIRunner(e, OPT, mode=mode, seed=9, **kw)


@pytest.mark.parametrize("case,use_graph", [("vel_mf_joint", True), ("vel_lrmn_cond", True), ("phase_nb", False)])
def test_perf_mode_resume_is_bitwise(case, use_graph, tmp_path):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    e0, r0 = _runner(spec, "perf", use_graph=use_graph)
    r0.run_perf(20)
    want_p, want_l = e0.params.clone().cpu(), r0.perf_losses()
    e1, r1 = _runner(spec, "perf", use_graph=use_graph)
    r1.run_perf(10)
    path = str(tmp_path / "ckpt.npz")
    r1.save(path)
    sd = r1.state_dict()
    assert sd["t"] == 10 and sd["step_idx"] == 10 and len(sd["losses"]) == 10
    e1.close()
    e2, r2 = _runner(spec, "perf", use_graph=use_graph)
    r2.load(path)
    r2.run_perf(10)
    nz = lambda t: torch.nan_to_num(t, neginf=-1e30)
    assert torch.equal(nz(e2.params.cpu()), nz(want_p))
    assert r2.perf_losses() == want_l and len(want_l) == 20
    assert torch.equal(r2.opt.m.cpu(), r0.opt.m.cpu()) and torch.equal(r2.opt.v.cpu(), r0.opt.v.cpu())
    # a checkpoint of another layout is refused
    other = H.spec_from_fixture(H.load_fixture(f"{H.GOLDEN}/ref_step_vel_mf_poisson.npz"))
    e3, r3 = _runner(other, "perf", use_graph=False)
    with pytest.raises(ValueError):
        r3.load(path)
    for e in (e0, e2, e3):
        e.close()


def test_parity_mode_resume_is_bitwise():
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_vel_lrmn_cond.npz")
    spec = H.spec_from_fixture(z)
    e0, r0 = _runner(spec, "parity")
    l0 = [r0.step() for _ in range(12)]
    e1, r1 = _runner(spec, "parity")
    l1 = [r1.step() for _ in range(6)]
    sd = r1.state_dict()
    e2, r2 = _runner(spec, "parity")
    r2.load_state_dict(sd)
    l2 = [r2.step() for _ in range(6)]
    assert l1 + l2 == l0 and r2.losses == l0
    assert torch.equal(torch.nan_to_num(e2.params, neginf=-1e30), torch.nan_to_num(e0.params, neginf=-1e30))
    for e in (e0, e1, e2):
        e.close()


def test_run_perf_zero_steps_is_a_no_op():
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_vel_mf_joint.npz")
    spec = H.spec_from_fixture(z)
    e, r = _runner(spec, "perf", use_graph=True)
    p0 = e.params.clone()
    r.run_perf(0)
    torch.cuda.synchronize()
    assert torch.equal(e.params, p0) and r.step_idx == 0 and int(r.step_dev.item()) == 0 and r.perf_losses() == []
    r.run_perf(3)
    assert r.step_idx == 3 and len(r.perf_losses()) == 3 and r.state_dict()["t"] == 3
    e.close()


def test_device_side_failure_latch():
    """C-ABI failure detection (reference: pyro.util.warn_if_nan inside SVI.step): the last kernel of a
    step latches the first step whose loss is NaN / Inf; vc_get_status reports it without a per-step host round trip."""
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_vel_mf_joint.npz")
    spec = H.spec_from_fixture(z)
    e, r = _runner(spec, "perf", use_graph=True)
    r.run_perf(5)
    assert e.status() == (True, -1, 0)
    e.view(e.params, "ν_locs")[0, 0] = float("nan")
    r.invalidate()                 # params edited behind the runner's back: the next step re-draws its sample from them
    r.run_perf(3)
    ok, first, n = e.status()
    assert not ok and first == 5 and n == 3
    assert "non-finite loss" in e.lib.vc_last_error(e._h).decode()
    assert not np.isfinite(r.perf_losses()[5])
    e.clear_status()
    assert e.status()[0] is True
    e.close()
