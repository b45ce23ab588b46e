"""Worker of tests/test_hip_fit_sharded.py: the tutorials' two-stage flow (phase fit -> conditioned two-sample velocity
fit, BASELINE.json configs[4]) through the PUBLIC entry points, run either as one process or as the ranks of a
torch.distributed job with the cells sharded.  Every rank builds the same full-size inputs; rank 0 writes the results.

  python tests/fit_shard_worker.py OUT.npz MODE        MODE = perf | parity
  python -m torch.distributed.run --nproc-per-node 2 ... tests/fit_shard_worker.py OUT.npz MODE

Test hook VC_BENCH_ONE_DEVICE=1: every rank on cuda:0 and gloo instead of RCCL (a 1-GPU box cannot host two RCCL ranks)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    one_device = os.environ.get("VC_BENCH_ONE_DEVICE", "0") == "1"
    device = torch.device("cuda:0" if one_device else f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

    from velocycle_amd import containers as C, preprocessing as P
    from velocycle_amd import pyro_compat as pyro
    from velocycle_amd.anndata_lite import AnnDataLite
    from velocycle_amd.fit_models import PhaseFitModel, VelocityFitModel
    from velocycle_amd.optim import ClippedAdam
    from velocycle_amd.workloads import make_velocity_spec

    ncell, ngene, n1, n2 = 1501, 150, 30, 30            # 2 samples x 1501 cells: odd total -> unequal shards
    sp = make_velocity_spec(ncell, ngene, "vjoint", n_conditions=2, Hw=0, seed=12)
    ad = AnnDataLite(sp.S.t().numpy(), sp.U.t().numpy())
    ad.obs["batch"] = [f"d{int(b)}" for b in sp.truth["batch"]]
    cyc = C.Cycle.from_array(sp.mu_nu.T.numpy(), sp.sd_nu.T.numpy(), list(ad.var.index))
    ph = C.Phases.from_array(sp.phixy_prior.T.numpy(), cell_names=list(ad.obs.index))
    Db = P.make_design_matrix(ad, ids="batch")
    opt = lambda n: ClippedAdam({"lr": 0.03, "lrd": (0.005 / 0.03) ** (1 / n), "betas": (0.80, 0.99)})
    res = {}

    torch.manual_seed(100)
    pyro.clear_param_store()
    mp = P.preprocess_for_phase_estimation(ad, cyc, ph, Db, n_harmonics=1)
    PhaseFitModel._default_elbo_fresh = True
    pf = PhaseFitModel(mp, num_samples=4, n_per_bin=2)
    pf.fit(opt(n1), num_steps=n1, verbose=False, mode=mode, seed=21)
    assert pf.engine.world_size == world and pf.engine.Nc_local < mp.Nc or world == 1
    res["phase_losses"] = np.array(pf.losses)
    for a in ("phis_pyro", "fourier_coef", "fourier_coef_sd", "disp_pyro", "delta_nus"):
        res["phase_" + a] = np.asarray(getattr(pf, a))
    for k in ("ϕxy", "ϕ", "ν", "ElogS", "ElogS2"):
        res["phase_post_" + k] = pf.posterior[k].numpy()

    cond = {"ϕxy": pf.phase_pyro.phi_xy_tensor.T, "ν": pf.cycle_pyro.means_tensor.T.unsqueeze(-2),
            "Δν": torch.tensor(pf.delta_nus), "shape_inv": torch.tensor(pf.disp_pyro).unsqueeze(-1)}
    spd = C.AngularSpeed.trivial_prior(condition_names=["d0", "d1"], harmonics=0)
    pyro.clear_param_store()
    mv = P.preprocess_for_velocity_estimation(ad, pf.cycle_pyro, pf.phase_pyro, spd, Db.float(), Db.float(), n_harmonics=1,
                                              count_factor=mp.count_factor, ω_n_harmonics=0, condition_on=cond)
    VelocityFitModel._default_elbo_fresh = True
    vf = VelocityFitModel(mv, condition_on=cond, num_samples=4, n_per_bin=2)
    vf.fit(opt(n2), num_steps=n2, verbose=False, mode=mode, seed=22)
    res["vel_losses"] = np.array(vf.losses)
    res["vel_kernel"] = np.array(vf.engine.stats["main_kernel"])
    for a in ("phis_pyro", "fourier_coef", "disp_pyro", "log_betas", "log_gammas", "velocity_coef"):
        res["vel_" + a] = np.asarray(getattr(vf, a))
    res["vel_loc"] = pyro.param("loc").numpy()
    for k in ("ω", "ϕ", "logβg", "νω", "ElogS", "ElogU", "ElogU2"):
        res["vel_post_" + k] = vf.posterior[k].numpy()
    res["world"] = np.array(world)
    res["nc_local"] = np.array(vf.engine.Nc_local)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        np.savez(out_path, **res)


if __name__ == "__main__":
    main()
