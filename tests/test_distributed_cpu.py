"""CPU, world_size 2 over gloo: the data-parallel host logic of SVIRunner (cell sharding, eps slicing,
ONE all-reduce of [loss hi/lo + replicated gradients], replicated ClippedAdam) with the oracle standing
in for the HIP engine as the local ELBO/gradient provider (test double; the product has no such path)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import velocycle_oracle as orc          # noqa: E402
from tests import helpers as H                      # noqa: E402


class OracleShardEngine:
    """Same surface as velocycle_amd.engine.HipEngine (the part SVIRunner touches), computed by the oracle
    on this rank's cells; rank > 0 drops the replicated prior/entropy terms exactly like the HIP engine."""

    def __init__(self, spec, rank, world_size):
        from velocycle_amd.engine import shard_bounds
        self.spec, self.rank, self.world_size = spec, rank, world_size
        self.device = torch.device("cpu")
        self.c0, self.c1 = shard_bounds(spec.Nc, rank, world_size)
        self.Nc_local = self.c1 - self.c0
        sl = slice(self.c0, self.c1)
        kw = {k: v for k, v in spec.__dict__.items() if k not in ("truth", "S_csr", "U_csr")}
        for k in ("S", "U", "Db", "D"):
            kw[k] = kw[k][:, sl].double() if kw[k] is not None else None
        for k in ("count_factor", "phixy_prior"):
            kw[k] = kw[k][sl].double()
        for k, v in kw.items():
            if isinstance(v, torch.Tensor) and v.dtype == torch.float32:
                kw[k] = v.double()
        kw["condition_on"] = {k: (v[sl] if k == "ϕxy" else v).double() for k, v in spec.condition_on.items()}
        self.p = orc.Problem(**kw)
        kw0 = dict(kw)
        for k in ("S", "U", "Db", "D"):
            kw0[k] = kw[k][:, :0] if kw[k] is not None else None
        for k in ("count_factor", "phixy_prior"):
            kw0[k] = kw[k][:0]
        kw0["condition_on"] = {k: (v[:0] if k == "ϕxy" else v) for k, v in kw["condition_on"].items()}
        self.p0 = orc.Problem(**kw0)
        par = orc.init_params(self.p)
        self.order = [k for k in par if k != "ϕxy_locs"] + ["ϕxy_locs"]
        self.shapes = {k: tuple(par[k].shape) for k in self.order}
        self.header = 4
        self.n_global = sum(par[k].numel() for k in self.order[:-1])
        self.n_local = par["ϕxy_locs"].numel()
        self.total = self.header + self.n_global + self.n_local
        self.params = torch.zeros(self.total, dtype=torch.float32)
        self.grad = torch.zeros(self.total, dtype=torch.float32)
        self.loss_dev = torch.zeros(1, dtype=torch.float64)

    def _unflatten(self, flat):
        out, off = {}, self.header
        for k in self.order:
            n = int(np.prod(self.shapes[k]))
            out[k] = flat[off:off + n].reshape(self.shapes[k]).double()
            off += n
        return out

    def init_params(self, cov=None):
        par = orc.init_params(self.p)
        off = self.header
        for k in self.order:
            n = par[k].numel()
            self.params[off:off + n] = par[k].reshape(-1).float()
            off += n
        return self.params

    def pack_eps(self, eps):
        e = {k: v.double() for k, v in eps.items() if not k.startswith("_")}
        e["ϕxy"] = e["ϕxy"][self.c0:self.c1]
        return e

    def elbo_grad(self, eps=None, step=0, **kw):
        leaves = {k: v.clone().requires_grad_(True) for k, v in self._unflatten(self.params).items()}
        loss, _, _ = orc.elbo_loss(self.p, leaves, eps)
        if self.rank > 0:      # replicated prior / entropy terms are counted once, on rank 0
            e0 = dict(eps)
            e0["ϕxy"] = eps["ϕxy"][:0]
            l0 = {k: (v[:0] if k == "ϕxy_locs" else v) for k, v in leaves.items()}
            loss = loss - orc.elbo_loss(self.p0, l0, e0)[0]
        loss.backward()
        off = self.header
        for k in self.order:
            g = leaves[k].grad
            n = leaves[k].numel()
            self.grad[off:off + n] = (torch.zeros(n) if g is None else g.reshape(-1)).float()
            off += n
        l = float(loss.detach())
        hi = np.float32(l)
        self.grad[0], self.grad[1] = float(hi), float(l - float(hi))
        self.loss_dev[0] = l


def _worker(rank, world, port, case, n_steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from velocycle_amd.svi import SVIRunner
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    eng = OracleShardEngine(spec, rank, world)
    run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.97, "betas": (0.8, 0.99)}, mode="parity", seed=5,
                    adam_impl="torch")
    losses = [run.step() for _ in range(n_steps)]
    q.put((rank, losses, eng.params[eng.header:eng.header + eng.n_global].clone().numpy(),
           eng.params[eng.header + eng.n_global:].clone().numpy(), (eng.c0, eng.c1)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["vel_mf_joint", "phase_nb"])
def test_two_rank_svi_equals_single_process(case):
    n_steps, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process oracle fit on the whole data set, same seed
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    p64 = H.problem_from_fixture(z)
    losses, par = orc.fit(p64, {"lr": 0.03, "lrd": 0.97, "betas": (0.8, 0.99)}, n_steps, seed=5)
    for r in res:
        assert np.allclose(r[1], losses, rtol=2e-6), (r[1], losses)        # every rank sees the global loss
    assert np.array_equal(res[0][2], res[1][2])                           # replicated parameters stay in sync
    order = [k for k in par if k != "ϕxy_locs"]
    flat = np.concatenate([par[k].reshape(-1).numpy() for k in order])
    fin = np.isfinite(flat)
    assert np.allclose(res[0][2][fin], flat[fin], rtol=1e-4, atol=1e-5)
    xy = np.concatenate([r[3] for r in res]).reshape(-1, 2)
    assert np.allclose(xy, par["ϕxy_locs"].numpy(), rtol=1e-4, atol=1e-5)
    assert res[0][4][1] == res[1][4][0] and res[1][4][1] == p64.Nc       # contiguous, complete shards


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from velocycle_amd.distributed import broadcast_int, dist_context, gather_cells
    from velocycle_amd.engine import shard_bounds
    Nc = 11
    sizes = [b - a for a, b in (shard_bounds(Nc, r, world) for r in range(world))]
    c0, c1 = shard_bounds(Nc, rank, world)
    full = torch.arange(3 * Nc * 2, dtype=torch.float32).reshape(3, Nc, 2)       # (draws, cells, xy)
    got = gather_cells(full[:, c0:c1], 1, sizes)
    rows = gather_cells(full[0, c0:c1], 0, sizes)
    seed = broadcast_int(1234 + rank)
    q.put((rank, dist_context()[:2], torch.equal(got, full), torch.equal(rows, full[0]), seed, sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_cells_and_seed_agreement_two_ranks():
    """The once-per-fit exchanges of the sharded fit(): unequal shards (6 + 5 cells) gathered along any axis give the
    full array on every rank; rank 0's seed reaches every rank."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, (rank, ctxinfo, ok3, ok2, seed, sizes) in enumerate(res):
        assert ctxinfo == (r, 2) and ok3 and ok2 and seed == 1234 and sizes == [6, 5]


def test_gather_cells_without_process_group_is_identity():
    from velocycle_amd.distributed import broadcast_int, dist_context, gather_cells
    assert dist_context() == (0, 1, None)
    x = torch.arange(6.).reshape(2, 3)
    assert torch.equal(gather_cells(x, 1, [3]), x) and broadcast_int(7) == 7


# ---------------------------------------------------------------------------------------------------------------------
# The sharded FUSED step's host loop (SVIRunner adam_impl="sharded", exchange="torch": phase A -> all_reduce of ONE exchange
# buffer -> phase B) over gloo with a test double that models the device protocol of vc_svi_run_sharded: phase A leaves this
# rank's gradient partial of the replicated parameters and its loss partial (four floats on fixed grids) in the buffer and
# updates its own cells' parameters; phase B applies ClippedAdam to the replicated parameters from the SUMMED buffer and files
# the summed loss.  The product has no such CPU path (HipEngine raises without a GPU); this covers rank / shard / exchange
# plumbing where a GPU is not available.
# ---------------------------------------------------------------------------------------------------------------------
def _split4(v):
    p0 = np.rint(v / 1048576.0) * 1048576.0
    r0 = v - p0
    p1 = np.rint(r0 * 0.5) * 2.0
    r1 = r0 - p1
    p2 = np.rint(r1 * 131072.0) / 131072.0
    return [np.float32(p0), np.float32(p1), np.float32(p2), np.float32(r1 - p2)]


class OracleShardedStepEngine(OracleShardEngine):
    def __init__(self, spec, rank, world_size):
        super().__init__(spec, rank, world_size)
        self.pfull = H.problem_from_spec(spec, torch.float64)

    def exchange_size(self):
        return self.header + self.n_global + 4

    def _eps(self, seed, step):
        g = torch.Generator().manual_seed(int(seed) * 100003 + int(step))
        return self.pack_eps(orc.draw_eps(self.pfull, g))

    def svi_run_sharded(self, xbuf, m, v, lr, lrd, b1, b2, adam_eps, clip, seed, step_dev, loss_buf=None, prime=False,
                        phase=3, n_steps=1):
        assert phase in (1, 2) and n_steps == 1
        ng = self.header + self.n_global
        t = int(step_dev.item()) + (1 if phase == 1 else 0)          # 1-based optimiser step of this SVI step
        step_size = lr * lrd ** t * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)

        def adam(sl):
            g = self.grad[sl].clamp(-clip, clip)
            mm, vv = m[sl.start - self.header:sl.stop - self.header], v[sl.start - self.header:sl.stop - self.header]
            mm.mul_(b1).add_(g, alpha=1 - b1)
            vv.mul_(b2).addcmul_(g, g, value=1 - b2)
            self.params[sl] -= step_size * mm / (vv.sqrt() + adam_eps)
        if phase == 1:
            self.elbo_grad(eps=self._eps(seed, t - 1))
            xbuf.zero_()
            xbuf[self.header:ng] = self.grad[self.header:ng]
            xbuf[ng:ng + 4] = torch.tensor(_split4(float(self.loss_dev[0])))
            adam(slice(ng, self.total))                               # rank-local parameters: no exchange needed
            step_dev += 1
        else:
            self.grad[self.header:ng] = xbuf[self.header:ng]          # the sum over ranks
            adam(slice(self.header, ng))
            loss_buf[(t - 1) % loss_buf.numel()] = float(xbuf[ng:ng + 4].double().sum())


def _sharded_worker(rank, world, port, case, n_steps, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from velocycle_amd.svi import SVIRunner
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    eng = OracleShardedStepEngine(H.spec_from_fixture(z), rank, world)
    run = SVIRunner(eng, {"lr": 0.03, "lrd": 0.97, "betas": (0.8, 0.99)}, mode="perf", seed=5)
    assert run.adam_impl == "sharded" and run.exchange == "torch" and not run.use_graph and run.xbuf.numel() == eng.exchange_size()
    run.run_perf(n_steps)
    q.put((rank, run.perf_losses(), eng.params[eng.header:eng.header + eng.n_global].clone().numpy(),
           eng.params[eng.header + eng.n_global:].clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["vel_mf_joint", "phase_nb"])
def test_two_rank_sharded_fused_host_loop_equals_single_process(case):
    n_steps, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, case, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    p64 = H.problem_from_fixture(z)
    eps_list = []
    for t in range(n_steps):
        g = torch.Generator().manual_seed(5 * 100003 + t)
        eps_list.append(orc.draw_eps(p64, g))
    losses, par = orc.fit(p64, {"lr": 0.03, "lrd": 0.97, "betas": (0.8, 0.99)}, n_steps, eps_list=eps_list,
                          params=orc.init_params(p64))
    for r in res:
        assert len(r[1]) == n_steps and np.allclose(r[1], losses, rtol=2e-6), (r[1], losses)
    assert np.array_equal(res[0][2], res[1][2])                           # replicated parameters stay in sync
    order = [k for k in par if k != "ϕxy_locs"]
    flat = np.concatenate([par[k].reshape(-1).numpy() for k in order])
    fin = np.isfinite(flat)
    assert np.allclose(res[0][2][fin], flat[fin], rtol=1e-4, atol=1e-5)
    xy = np.concatenate([r[3] for r in res]).reshape(-1, 2)
    assert np.allclose(xy, par["ϕxy_locs"].numpy(), rtol=1e-4, atol=1e-5)
