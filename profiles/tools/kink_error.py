#!/usr/bin/env python
"""Where the float32 gradient error of a step comes from, and whether the HIP kernels carry MORE of it than the reference's own
float32 arithmetic (a `loc` block at 3.8e-3 of its max-norm against 1.2e-3 for torch float32).

For several draws of (params, eps) on the two-sample configuration (2 x 5 000 cells x 500 genes; V-joint, tutorial flow with the LRMN
guide, tutorial flow with the mean-field guide) every gradient block is evaluated three ways -- HIP (float32), the oracle in float32
(= the reference's arithmetic, op by op), the oracle in float64 (the checker) -- and the script prints per block
    err_hip / max-norm, err_f32 / max-norm, their ratio, the share of elements within 1e-3 of their own magnitude,
and the census of the relu kink of ElogU (velocity_inference_model.py:365-368): z = nu . zeta'(phi) omega + gamma enters as
log(relu(z) + 1e-5), so d / dz carries 1 / (z + 1e-5); an element with 0 < z < 1e-4 turns a float32 rounding of z (~1e-7) into a
0.1-1 % error of ITS term, and that one term can be as large as the whole block's max-norm.

    python profiles/tools/kink_error.py [n_draws] > profiles/r06_kink_error.txt          (needs a GPU)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import velocycle_oracle as orc  # noqa: E402
from tests import helpers as H  # noqa: E402
from velocycle_amd.engine import HipEngine  # noqa: E402
from velocycle_amd.rng import draw_eps  # noqa: E402
from velocycle_amd.workloads import make_velocity_spec  # noqa: E402


def main():
    n_draws = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ratios = {}
    for mode in ("vjoint", "vcond", "vcond_mf"):
        spec = make_velocity_spec(5000, 500, mode, n_conditions=2, Hw=1, seed=6)
        p64 = H.problem_from_spec(spec, torch.float64)
        p32 = p64.to(torch.float32)
        eng = HipEngine(spec)
        for draw in range(n_draws):
            g = torch.Generator().manual_seed(100 + draw)
            first = draw_eps(spec, g)
            eps = draw_eps(spec, g)
            eng.init_params(first.get("_cov_factor_draw"))
            # move the parameters off their initial values (a fit is not at its initialisation): one pseudo-step of noise
            gp = torch.Generator().manual_seed(200 + draw)
            with torch.no_grad():
                noise = 0.05 * torch.randn(eng.params.shape, generator=gp).to(eng.device)
                fin = torch.isfinite(eng.params)
                eng.params[fin] += noise[fin]
            eng.elbo_grad(eps=eng.pack_eps(eps))
            torch.cuda.synchronize()
            par = {n: v.detach().cpu().double() for n, v in eng.named().items()}
            e64 = {k: v.double() for k, v in eps.items() if not k.startswith("_")}
            l64, g64, val, det = orc.loss_and_grads(p64, par, e64)
            _, g32, _, _ = orc.loss_and_grads(p32, {k: v.float() for k, v in par.items()}, {k: v.float() for k, v in e64.items()})
            # kink census from the float64 sites
            nu = val["ν"].reshape(spec.Ng, -1)
            z = (nu @ det["ζ_dϕ"].reshape(spec.Nc, -1).T) * det["ω"].reshape(1, -1) + det["γg"].reshape(-1, 1)
            zz = z.numpy()
            census = {t: int(((zz > 0) & (zz < t)).sum()) for t in (1e-5, 1e-4, 1e-3)}
            print(f"== {mode} draw {draw}: loss rel err {abs(eng.loss() - l64) / abs(l64):.1e}; elements with 0 < z < 1e-5 / 1e-4 / 1e-3: "
                  f"{census[1e-5]} / {census[1e-4]} / {census[1e-3]} of {zz.size}; z <= 0 (relu off): {int((zz <= 0).sum())}")
            for name, got in eng.named(eng.grad).items():
                want = g64[name].numpy().reshape(-1)
                fin = np.isfinite(want)
                if not fin.any() or np.abs(want[fin]).max() == 0:
                    continue
                gh = got.cpu().numpy().astype(np.float64).reshape(-1)[fin]
                t32 = g32[name].numpy().astype(np.float64).reshape(-1)[fin]
                w = want[fin]
                sc = np.abs(w).max()
                eh, e3 = np.abs(gh - w), np.abs(t32 - w)
                sh_h = float((eh <= 1e-3 * np.abs(w) + 1e-6 * sc).mean())
                sh_3 = float((e3 <= 1e-3 * np.abs(w) + 1e-6 * sc).mean())
                r = eh.max() / max(e3.max(), 1e-300)
                ratios.setdefault((mode, name), []).append((eh.max() / sc, e3.max() / sc, r, sh_h, sh_3))
                print(f"   {name:14s} hip {eh.max() / sc:.2e}  float32 oracle {e3.max() / sc:.2e}  ratio {r:5.2f}   within 1e-3 of itself: hip {sh_h:.4f} float32 oracle {sh_3:.4f}"
                      + ("   <-- beyond 2e-3" if eh.max() / sc > 2e-3 else ""))
        eng.close()
    print("\n== summary over draws: per (mode, block) median / max of err_hip / err_float32-oracle, worst err_hip, worst err_f32, min shares")
    allr = []
    for (mode, name), rows in ratios.items():
        a = np.array(rows)
        allr += [x[2] for x in rows if max(x[0], x[1]) > 2e-4]       # only where an error is visible at all
        print(f"   {mode:9s} {name:14s} ratio median {np.median(a[:, 2]):5.2f} max {a[:, 2].max():5.2f}   worst hip {a[:, 0].max():.2e}  worst float32 {a[:, 1].max():.2e}"
              f"   min share hip {a[:, 3].min():.4f} float32 {a[:, 4].min():.4f}")
    allr = np.array(allr)
    print(f"\nblocks x draws with a visible error (> 2e-4 of the max-norm on either side): {allr.size}; ratio hip / float32 oracle: "
          f"median {np.median(allr):.2f}, 10 % / 90 % quantiles {np.quantile(allr, 0.1):.2f} / {np.quantile(allr, 0.9):.2f}, max {allr.max():.2f}")


if __name__ == "__main__":
    main()
