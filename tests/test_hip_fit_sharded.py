"""GPU, two real processes: `PhaseFitModel.fit` / `VelocityFitModel.fit` with the cells sharded over the ranks of a
torch.distributed job (BASELINE.json configs[4] = two-sample flow on 2 GPUs) against the same flow in
one process.  On a 1-GPU box both ranks sit on cuda:0 and exchange through gloo (VC_BENCH_ONE_DEVICE hook); everything
else -- shard bounds, seed agreement, the per-step all-reduce, the final gathers of ϕxy_locs / per-cell posterior sites /
ElogS, ElogU columns -- is the path that runs over RCCL on a multi-GPU node.  Shard-count invariance: losses to 1e-6,
attributes and posterior summaries to 1e-3 (float32 reassociation of the two partial sums through 30 Adam steps)."""
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
    for k in ("phase_losses", "vel_losses"):
        assert len(a[k]) == 30
        assert np.allclose(a[k], b[k], rtol=1e-6, atol=0), (k, np.abs(a[k] / b[k] - 1).max())
    for k in a.files:
        if k in ("world", "nc_local", "vel_kernel", "phase_losses", "vel_losses"):
            continue
        assert a[k].shape == b[k].shape, (k, a[k].shape, b[k].shape)
        fin = np.isfinite(a[k])
        assert np.array_equal(fin, np.isfinite(b[k])), k
        assert np.allclose(a[k][fin], b[k][fin], rtol=1e-3, atol=1e-3), (k, np.abs(a[k][fin] - b[k][fin]).max())
    # the gathered per-cell results really cover every cell of both samples
    assert a["vel_post_ω"].shape[-1] == 3002 and b["phase_phis_pyro"].shape == (2, 3002)
