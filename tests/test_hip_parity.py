"""GPU parity: the HIP path (through the C ABI) against the float64 oracle on the golden fixtures.

Tolerances (float32 kernels with hardware exp/log/rcp vs float64 oracle, stated per north_star):
loss 1e-5 relative; every gradient block 2e-3 of its max-norm (the clamp-free, pre-optimiser gradient).
"""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _engine(spec, **kw):
    from velocycle_amd.engine import HipEngine
    return HipEngine(spec, **kw)


@pytest.mark.parametrize("case", H.STEP_CASES)
def test_single_step_matches_oracle(case):
    z = H.load_fixture(f"{H.GOLDEN}/ref_step_{case}.npz")
    spec = H.spec_from_fixture(z)
    eng = _engine(spec)
    eng.set_params({k[4:]: torch.tensor(v) for k, v in z.items() if k.startswith("par_")})
    eps = eng.pack_eps({k[4:]: torch.tensor(v) for k, v in z.items() if k.startswith("eps_")})
    eng.elbo_grad(eps=eps)
    torch.cuda.synchronize()
    loss = eng.loss()
    assert abs(loss - float(z["loss64"])) <= 1e-5 * abs(float(z["loss64"])), (loss, float(z["loss64"]))
    hdr = eng.grad[:2].double().sum().item()
    assert abs(hdr - loss) <= 1e-6 * abs(loss)
    g = {k: v.cpu().numpy() for k, v in eng.named(eng.grad).items()}
    for name, got in g.items():
        want = z["grad64_" + name]
        fin = np.isfinite(want)
        err = np.abs(got[fin] - want[fin]).max() if fin.any() else 0.0
        tol = 2e-3 * max(np.abs(want[fin]).max() if fin.any() else 0.0, 1e-3)
        assert err <= tol, f"{case}: grad {name}: max err {err} > {tol}"
    # sites agree as well
    for k, v in z.items():
        if k.startswith("val64_") and k[6:] in ("ν", "logγg", "logβg", "νω", "shape_inv", "ϕxy"):
            got = eng.read_site(k[6:]).numpy().reshape(v.shape)
            assert np.allclose(got, v, rtol=1e-5, atol=1e-5), f"{case}: site {k[6:]}"
    eng.close()
