"""GPU, BASELINE.json's full size (50k cells x 2k genes, velocity inference): size-independent properties of
the hot path -- shard additivity, layout (genes-per-lane) invariance, bitwise determinism -- plus the float64
oracle on a 2 000-cell slice of the same data."""
import numpy as np
import pytest
import torch

from oracle import velocycle_oracle as orc

pytestmark = pytest.mark.gpu

NC, NG = 50000, 2000


@pytest.fixture(scope="module")
def spec():
    from velocycle_amd.workloads import make_velocity_spec
    return make_velocity_spec(NC, NG, "vjoint", n_conditions=1, Hw=1, seed=0, device="cuda")


def _run(spec, seed=5, step=2, **kw):      # kw: rank / world_size / tuning of the engine
    from velocycle_amd.engine import HipEngine
    e = HipEngine(spec, **kw)
    e.init_params()
    e.elbo_grad(eps=None, seed=seed, step=step)
    torch.cuda.synchronize()
    return e


def test_full_size_shard_additivity_and_determinism(spec):
    full = _run(spec)
    ref = full.grad[: full.header + full.n_global].double().cpu()
    loss = full.loss()
    # bitwise determinism of a repeated evaluation (fixed-order reductions, no float atomics)
    g0 = full.grad.clone()
    full.elbo_grad(eps=None, seed=5, step=2)
    torch.cuda.synchronize()
    assert torch.equal(g0, full.grad) and full.loss() == loss
    tot = torch.zeros_like(ref)
    xy = []
    for r in range(4):
        s = _run(spec, rank=r, world_size=4)
        tot += s.grad[: s.header + s.n_global].double().cpu()
        xy.append(s.view(s.grad, "ϕxy_locs").cpu())
        s.close()
    assert abs((tot[0] + tot[1]) - loss) <= 2e-7 * abs(loss)
    scale = ref[4:].abs().max()
    assert (tot[4:] - ref[4:]).abs().max() <= 2e-5 * scale
    assert torch.allclose(torch.cat(xy), full.view(full.grad, "ϕxy_locs").cpu(), rtol=1e-4, atol=1e-3)
    full.close()


def test_full_size_layout_invariance(spec):
    from velocycle_amd.tuning import Tuning
    res = []
    for gpl in (4, 8):
        e = _run(spec, tuning=Tuning(genes_per_lane=gpl))
        assert f"gpl{gpl}" in e.stats["main_kernel"]
        res.append((e.loss(), e.grad.double().cpu()))
        e.close()
    assert abs(res[0][0] - res[1][0]) <= 1e-7 * abs(res[0][0])
    d = (res[0][1][4:] - res[1][1][4:]).abs().max()
    assert d <= 2e-5 * res[0][1][4:].abs().max()


def test_oracle_on_a_slice_of_the_full_problem(spec):
    """First 2 000 cells x all 2 000 genes of the benchmark data against the float64 oracle."""
    from velocycle_amd.engine import HipEngine
    from velocycle_amd.spec import ModelSpec
    n = 2000
    kw = {k: v for k, v in spec.__dict__.items() if k not in ("truth", "S_csr", "U_csr")}
    for k in ("S", "U", "D", "Db"):
        kw[k] = kw[k][:, :n].cpu().contiguous()
    for k in ("count_factor", "phixy_prior"):
        kw[k] = kw[k][:n].cpu()
    sub = ModelSpec(**kw)
    e = HipEngine(sub)
    e.init_params()
    e.elbo_grad(eps=None, seed=3, step=0)
    torch.cuda.synchronize()
    eps_flat = e.read_site("eps")
    p = orc.Problem(**{k: (v.double() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()})
    shapes = {"ν": (p.Ng, p.Nh), "νω": (p.Nx, p.Nhw), "ϕxy": (p.Nc, 2)}
    eps = {k: eps_flat[o:o + s].double().reshape(shapes.get(k, (s,))) for k, (o, s) in e.eps_slices.items()}
    par = {k: v.detach().cpu().double() for k, v in e.named().items()}
    l64, g64, _, _ = orc.loss_and_grads(p, par, eps)
    assert abs(e.loss() - l64) <= 1e-5 * abs(l64)
    for name, got in e.named(e.grad).items():
        want = g64[name].numpy()
        err = np.abs(got.cpu().numpy() - want).max()
        assert err <= 3e-3 * max(np.abs(want).max(), 1e-3), (name, err)
    e.close()
