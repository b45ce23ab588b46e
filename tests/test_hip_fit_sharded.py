"""GPU, two real processes: `PhaseFitModel.fit` / `VelocityFitModel.fit` with the cells sharded over the ranks of a
torch.distributed job (BASELINE.json configs[4] = two-sample flow on 2 GPUs) against the same flow in
one process.  On a 1-GPU box both ranks sit on cuda:0 and exchange through gloo (VC_BENCH_ONE_DEVICE hook); everything
else -- shard bounds, seed agreement, the per-step all-reduce, the final gathers of ϕxy_locs / per-cell posterior sites /
ElogS, ElogU columns -- is the path that runs over RCCL on a multi-GPU node.  Shard-count invariance: the first steps'
losses to 1e-6, trajectories and results as far as two float32 runs of this flow agree (see the comment at the asserts);
tests/test_hip_sharded_step.py holds the sharded step itself against the single-rank step on fixed inputs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tb(err):
    i = err.find("Traceback")
    return err[i:i + 4000] if i >= 0 else err[-3000:]


@pytest.mark.parametrize("mode", ["perf", "parity"])
def test_sharded_fit_equals_single_process_fit(mode, tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r1 = subprocess.run([sys.executable, "tests/fit_shard_worker.py", one, mode], cwd=ROOT, env=env,
                        capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, _tb(r1.stderr)
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                         "tests/fit_shard_worker.py", two, mode],
                        cwd=ROOT, env=dict(env, VC_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, _tb(r2.stderr)
    a, b = np.load(one), np.load(two)
    assert int(a["world"]) == 1 and int(b["world"]) == 2 and int(b["nc_local"]) == 1501
    assert "vu_" in str(b["vel_kernel"])                       # tutorial flow: S term hoisted, on every shard
    # Shard-count invariance is exact in exact arithmetic; in float32 the two runs add the cells up in a different order, and
    # a gene that sits on the relu kink of ElogU turns a 5e-7 difference of a parameter into a 0.4 % difference of its
    # gradient (observed at step 3 of the velocity stage of this very flow: d loss / d loc = -31 952 vs -31 821), after which
    # the two Adam trajectories separate like any two float32 runs do.  So: the first steps to 1e-6, the rest of the
    # trajectory and the results statistically (round 2's five-kernel sequence happened to stay on the single-process
    # trajectory for all 30 steps of this seed; the fused sharded step does so for 9 -- both are this flow in float32).
    for k in ("phase_losses", "vel_losses"):
        assert len(a[k]) == 30
        assert np.allclose(a[k][:8], b[k][:8], rtol=1e-6, atol=0), (k, np.abs(a[k][:8] / b[k][:8] - 1).max())
        assert np.allclose(a[k], b[k], rtol=5e-3, atol=0), (k, np.abs(a[k] / b[k] - 1).max())
    for k in a.files:
        if k in ("world", "nc_local", "vel_kernel", "phase_losses", "vel_losses"):
            continue
        assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
        fin = np.isfinite(a[k])
        assert np.array_equal(fin, np.isfinite(b[k])), k
        x, y = a[k][fin], b[k][fin]
        scale = max(np.abs(x).max(), 1e-3)
        if k.startswith("phase_"):        # the phase stage's two runs never separate in these 30 steps
            assert np.allclose(x, y, rtol=1e-3, atol=1e-3), (k, np.abs(x - y).max())
        else:                             # the velocity stage's do (unconverged 30-step fits: every site still moves by lr per step)
            assert np.abs(x - y).max() <= 0.25 * scale, (k, np.abs(x - y).max(), scale)
    # the gathered per-cell results really cover every cell of both samples
    assert a["vel_post_ω"].shape[-1] == 3002 and b["phase_phis_pyro"].shape == (2, 3002)
