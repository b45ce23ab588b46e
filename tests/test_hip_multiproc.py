"""GPU, two real processes: bench.py's N > 1 path end to end on a 1-GPU box.

torch.distributed.run starts two ranks; the test hook VC_BENCH_ONE_DEVICE=1 puts both on cuda:0 and exchanges the
gene-level gradient buffer through gloo instead of RCCL (RCCL refuses two ranks on one device).  Everything else is the
path the driver launches on 8 GPUs: rendezvous, cell sharding with offsets, Philox eps sliced per shard, the all-reduce
between the gradient kernels and the optimiser kernel, barrier + max-over-ranks timing, one JSON line from rank 0.
The loss trajectory must equal the single-process run (shard-count invariance across processes)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZE = ["--cells", "6000", "--genes", "300", "--steps", "30", "--warmup", "5", "--repeats", "1", "--roofline-launches", "20"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_two_process_bench_matches_single_process():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, "bench.py", *SIZE, "--no-cpu-baseline", "--no-extra-modes", "--no-graph"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = _json_line(one.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", *SIZE],
                         cwd=ROOT, env=dict(env, VC_BENCH_ONE_DEVICE="1"), capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    j2 = _json_line(two.stdout)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["steps"] == 30 and j2["scaling"] == "strong"
    # the fields round 4 added to the line: the step-level fraction next to the kernel-level one, and the OPT-IN measurement under
    # its own name (never `value`)
    r1 = j1["roofline"]
    assert 0 < r1["step_frac"] < r1["frac"] and r1["step_overhead_us"] > 0 and j1["config"]["launches_per_step"] in (2, 3)
    assert j1["opt_in_loss_every_10"]["steps_per_s"] > 0 and "opt_in_loss_every_10" not in j2
    assert "all-reduce" in j2["config"]["step"] and "cpu_baseline" not in j2
    assert j2["roofline"]["algorithmic_bytes_per_launch"] * 2 == j1["roofline"]["algorithmic_bytes_per_launch"]
    # first loss: the same parameters on both sides, only the order of the cell sums differs; the loss 34 steps later: two
    # float32 trajectories of this flow (a re-associated sum moves a gene on the relu kink of ElogU, helpers.py) -- the sharded
    # step itself is held against the single-rank step on fixed inputs in tests/test_hip_sharded_step.py
    for (a, b), tol in zip(zip(j1["loss_first_last"], j2["loss_first_last"]), (1e-6, 2e-4)):
        assert abs(a - b) <= tol * abs(a), (j1["loss_first_last"], j2["loss_first_last"])
    assert j1["loss_first_last"][1] < j1["loss_first_last"][0]


def test_bench_gpus_2_launches_itself():
    """`python bench.py --gpus 2 ...` as the driver writes it for N = 1 (no launcher in front): the parent, which has made no
    GPU call, starts torch.distributed.run as a child, relays rank 0's ONE JSON line on stdout and the exit code."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VC_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *SIZE], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    out_lines = [l for l in two.stdout.splitlines() if l.strip()]
    assert len(out_lines) == 1, two.stdout[-2000:]
    j = json.loads(out_lines[0])
    assert j["n_gpus"] == 2 and j["distributed"]["world_size"] == 2 and j["steps"] == 30
    # the weak-scaling block of the same invocation: --cells cells per rank, the same kernels and exchange
    w = j["weak"]
    assert w["scaling"] == "weak" and w["cells_per_rank"] == 6000 and w["cells_total"] == 12000 and w["value"] > 0
    assert w["exchange"] == j["distributed"]["exchange"] and len(w["repeat_ms_per_step"]) >= 5


def test_a_stuck_multi_rank_run_is_given_up():
    """A rank that cannot finish (here: a deadline shorter than start-up) ends itself with exit code 124 and the launcher
    ends the others: an N > 1 bench run never hangs its caller for longer than VC_BENCH_DEADLINE_S (default 900 s)."""
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VC_BENCH_ONE_DEVICE="1", VC_BENCH_DEADLINE_S="0.2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    two = subprocess.run([sys.executable, "bench.py", "--gpus", "2", *SIZE], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=300)
    assert two.returncode != 0 and time.time() - t0 < 200
    assert "is stuck; exit 124" in two.stderr, two.stderr[-2000:]
    assert not [l for l in two.stdout.splitlines() if l.startswith('{"metric"')]


def _two_ranks(env_extra, extra_args=(), n=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VC_BENCH_ONE_DEVICE="1", **env_extra)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", str(n), *SIZE, *extra_args],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return _json_line(r.stdout)


@pytest.mark.parametrize("fold", ["1", "0", "x"])
def test_two_process_p2p_exchange_equals_the_collective(fold):
    """(fold = x: phases A and B in ONE launch with the exchange at block granularity, vc_tail_x_kernel -- K_main + one launch per step;
    fold = 1, the default since round 6: phase B runs the exchange's publish / wait protocol itself and adds the ranks' slots where it
    reads them -- K_main, phase A, phase B and nothing else; fold = 0: the exchange as a launch of its own.)  VC_EXCHANGE=p2p: the one-shot exchange over peer-mapped device memory (vc_p2p_exchange.hip) between two PROCESSES (both
    on cuda:0 -- hipIpc works between processes on one device, which is all a 1-GPU box can offer): region export / import,
    step-stamped flags, double-buffered slots, fixed-rank-order sum, all enqueued from one C call per run.  Same losses as the
    gloo all-reduce of the same two ranks (two addends: a + b is b + a, bit for bit), no exchange time-out latched."""
    ref = _two_ranks({"VC_EXCHANGE": "torch"})
    got = _two_ranks({"VC_EXCHANGE": "p2p", "VC_P2P_TIMEOUT_S": "20", "VC_P2P_FOLD": "1" if fold == "x" else fold,
                      "VC_P2P_ONE_LAUNCH": "1" if fold == "x" else "0"})
    assert got["distributed"]["exchange"] == "p2p" and ref["distributed"]["exchange"] == "torch"
    assert got["nonfinite_loss_steps"] == 0
    for a, b in zip(ref["loss_first_last"], got["loss_first_last"]):
        assert abs(a - b) <= 1e-9 * abs(a), (ref["loss_first_last"], got["loss_first_last"])


@pytest.mark.parametrize("one_launch", ["0", "1"])
def test_four_process_p2p_exchange(one_launch):
    """The same with FOUR processes (unequal shards: 6 000 cells over 4 ranks of 1 500, tutorial-flow kernel): four regions,
    four flags per region, the sum in rank order 0..3 on every rank.  gloo adds the four buffers in another order, so the
    losses agree to float32 rounding of the re-associated sum instead of bit for bit."""
    ref = _two_ranks({"VC_EXCHANGE": "torch"}, ("--mode", "vcond"), n=4)
    got = _two_ranks({"VC_EXCHANGE": "p2p", "VC_P2P_TIMEOUT_S": "30", "VC_P2P_ONE_LAUNCH": one_launch}, ("--mode", "vcond"), n=4)
    assert got["distributed"]["exchange"] == "p2p" and got["distributed"]["world_size"] == 4 and got["nonfinite_loss_steps"] == 0
    for (a, b), tol in zip(zip(ref["loss_first_last"], got["loss_first_last"]), (2e-6, 2e-4)):      # first step | 34 steps on
        assert abs(a - b) <= tol * abs(a), (ref["loss_first_last"], got["loss_first_last"])


def test_p2p_exchange_survives_a_peer_that_stops():
    """A rank that stops publishing must not hang its peers' device: the exchange kernel's wait is bounded (VC_P2P_TIMEOUT_S),
    the step is poisoned (NaN loss -- never a partial sum) and vc_get_status reports VC_ERR_STATE (tests/p2p_dead_peer_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VC_P2P_TIMEOUT_S="1", VC_TEST_PEER_SLEEP_S="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "tests/p2p_dead_peer_worker.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("DEAD_PEER ")]
    assert len(lines) == 1, r.stdout[-2000:]
    got = json.loads(lines[0][len("DEAD_PEER "):])
    assert got["nan"] == [True, True], got                 # both lonely steps poisoned
    # ONE bounded wait of 1 s (the verdict is sticky: a rank that has poisoned a step publishes that and does not wait again --
    # ADVICE r4), not the peer's 8 s
    assert 0.8 <= got["waited_s"] < 7.0, got
    assert "peer-to-peer exchange" in got["status"] and "did not publish" in got["status"], got
