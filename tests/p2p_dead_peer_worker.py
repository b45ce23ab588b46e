"""Worker of tests/test_hip_multiproc.py::test_p2p_exchange_survives_a_peer_that_stops: two ranks on cuda:0 (gloo rendezvous,
hipIpc regions), the one-shot peer-to-peer exchange.  Both ranks run a few steps together; then rank 1 stops stepping (it stays
alive but never publishes again) and rank 0 steps on alone: its exchange kernel must give up within VC_P2P_TIMEOUT_S, poison
the step and latch VC_ERR_STATE -- not hang the device.  Rank 0 prints one line that the test checks.

  python -m torch.distributed.run --nproc-per-node 2 ... tests/p2p_dead_peer_worker.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    from velocycle_amd.engine import HipEngine, HipEngineError
    from velocycle_amd.tuning import Tuning
    from velocycle_amd.svi import SVIRunner
    from velocycle_amd.workloads import make_velocity_spec

    spec = make_velocity_spec(3000, 200, "vjoint", 1, 1, seed=3, device=device)
    eng = HipEngine(spec, device=device, rank=rank, world_size=world, tuning=Tuning.from_env())
    run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.9999, "betas": (0.8, 0.99)}, mode="perf", seed=5, exchange="p2p")
    assert run.exchange == "p2p"
    run.run_perf(6, sync=True)
    ok, _, _ = eng.status()
    together = run.loss_hist[:6].tolist()
    assert ok and all(x == x for x in together), together
    dist.barrier()
    if rank == 1:
        time.sleep(float(os.environ.get("VC_TEST_PEER_SLEEP_S", "8")))      # alive, mapped, silent
        dist.barrier()
        return
    t0 = time.time()
    run.run_perf(2, sync=True)               # the peer never raises its flag again
    waited = time.time() - t0
    alone = run.loss_hist[6:8].tolist()
    msg = ""
    try:
        eng.status()
    except HipEngineError as ex:
        msg = str(ex)
    import json
    print("DEAD_PEER " + json.dumps({"waited_s": round(waited, 2), "nan": [x != x for x in alone], "status": msg}), flush=True)
    dist.barrier()


if __name__ == "__main__":
    main()
