"""`HipEngine`: one ELBO + reparameterised-gradient evaluator on one MI355X, through the C ABI.

It replaces, for one data shard, what `Trace_ELBO(num_particles=1).loss_and_grads(model, guide, mp)`
does inside `svi.step` of the reference (phase_inference_model.py:169, velocity_inference_model.py:120).
PyTorch is used for device memory and streams only; every number is produced by the HIP kernels of
`velocycle_amd/csrc`.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _lib
from .spec import ModelSpec
from .tuning import Tuning


class HipEngineError(RuntimeError):
    pass


def shard_bounds(Nc: int, rank: int, world_size: int):
    """Contiguous cell blocks, sizes differing by at most one."""
    base, rem = divmod(Nc, world_size)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def _f32c(t, device="cpu"):
    return torch.as_tensor(t).detach().to(device=device, dtype=torch.float32).contiguous()


class HipEngine:
    def __init__(self, spec: ModelSpec, device: Optional[torch.device] = None, rank: int = 0,
                 world_size: int = 1, tuning: Optional[Tuning] = None):
        self.lib = _lib.load()
        self.tuning = tuning if tuning is not None else Tuning()
        if not torch.cuda.is_available():
            raise HipEngineError("velocycle_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self.spec, self.rank, self.world_size = spec, rank, world_size
        self.c0, self.c1 = shard_bounds(spec.Nc, rank, world_size)
        self.Nc_local = self.c1 - self.c0
        if self.Nc_local <= 0:
            raise HipEngineError("more ranks than cells")
        self._h = C.c_void_p()
        self._keep = []
        cfg = _lib.vc_config(
            abi_version=_lib.VC_ABI_VERSION, model=_lib.MODEL[spec.kind], guide=_lib.GUIDE[spec.guide],
            noise=self._noise_id(spec.noisemodel), with_delta_nu=int(spec.with_delta_nu),
            n_harmonics=spec.H, n_harmonics_w=spec.Hw, Nb=spec.Nb, Nx=spec.Nx, lrmn_rank=spec.rho_rank,
            rank=rank, world_size=world_size, Ng=spec.Ng, Nc_local=self.Nc_local, Nc_global=spec.Nc,
            cell_offset=self.c0, gamma_alpha=spec.gamma_alpha, gamma_beta=spec.gamma_beta,
            sigma_ln_s=spec.sigma_ln_s, sigma_ln_u=spec.sigma_ln_u, rho_mean=spec.rho_mean,
            rho_std=spec.rho_std, rho_scale=spec.rho_scale)
        rc = self.lib.vc_create(C.byref(cfg), C.byref(self._h))
        if rc != _lib.VC_OK:
            msg = self.lib.vc_last_error(None).decode()
            self._h = C.c_void_p()
            if rc == _lib.VC_ERR_UNSUPPORTED:
                raise NotImplementedError(msg)
            raise ValueError(msg)
        try:
            if hasattr(self.lib, "vc_set_tuning"):       # (absent only from an older build selected with VC_LIB_PATH + VC_LIB_OLDER)
                self._check(self.lib.vc_set_tuning(self._h, C.byref(self.tuning.to_c())))
            elif self.tuning != Tuning():
                raise HipEngineError("this build of the library has no vc_set_tuning: only the default Tuning can run on it")
            self._setup()
        except Exception:
            self.close()
            raise

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _noise_id(name):
        if name not in _lib.NOISE:
            raise ValueError(f"{name} not allowed")      # velocity_inference_model.py:388
        return _lib.NOISE[name]

    def _check(self, rc):
        if rc != _lib.VC_OK:
            msg = self.lib.vc_last_error(self._h).decode()
            if rc == _lib.VC_ERR_UNSUPPORTED:
                raise NotImplementedError(msg)
            if rc == _lib.VC_ERR_ARG:
                raise ValueError(msg)
            raise HipEngineError(msg)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _setup(self):
        sp, lib, h = self.spec, self.lib, self._h
        sl = slice(self.c0, self.c1)
        if sp.S_csr is not None and (sp.kind == "phase" or sp.U_csr is not None):
            # CSR ingest (cells x genes, e.g. AnnData layers): this rank's rows go to the device as they are and a
            # scatter kernel fills the blocked HBM layout; the dense matrices are never formed or uploaded
            import numpy as np
            for which, mat in ((0, sp.S_csr), (1, sp.U_csr if sp.kind == "velocity" else None)):
                if mat is None:
                    continue
                m = mat.tocsr()[self.c0:self.c1]
                m.sum_duplicates()
                indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
                indices = np.ascontiguousarray(m.indices, dtype=np.int32)
                data = np.ascontiguousarray(m.data, dtype=np.float32)
                if m.shape != (self.Nc_local, sp.Ng):
                    raise ValueError(f"CSR counts must be (cells, genes) = {(sp.Nc, sp.Ng)}")
                self._check(lib.vc_set_counts_csr(h, which, indptr.ctypes.data_as(C.c_void_p),
                                                  indices.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
                                                  int(m.nnz), 0))
        else:
            # counts: pass the caller's storage as it is (strided view, host or device); device memory is not copied
            S = sp.S[:, sl]
            U = sp.U[:, sl] if sp.U is not None else None
            for m in (S, U):
                if m is not None and m.dtype != torch.float32:
                    raise ValueError("count matrices must be float32 (the reference passes S.T.float())")
            on_dev = int(S.is_cuda)
            if U is not None and U.is_cuda != S.is_cuda:
                raise ValueError("S and U must live on the same device")
            norm = lambda m: m if (m is None or (m.stride(0) > 0 and m.stride(1) > 0)) else m.contiguous()
            S, U = norm(S), norm(U)
            if U is not None and U.stride() != S.stride():
                S, U = S.contiguous(), U.contiguous()
            gs, cs = S.stride()
            self._counts_keepalive = (S, U)          # read by vc_finalize below
            self._check(lib.vc_set_counts(h, C.c_void_p(S.data_ptr()),
                                          C.c_void_p(U.data_ptr()) if U is not None else None, gs, cs, on_dev))
        cf = _f32c(sp.count_factor.reshape(-1)[sl])
        D = _f32c(sp.D[:, sl]) if sp.D is not None else None
        Db = _f32c(sp.Db[:, sl]) if (sp.with_delta_nu and sp.Nb > 0) else None
        pxy = _f32c(sp.phixy_prior[sl])
        self._check(lib.vc_set_cell_data(h, C.c_void_p(cf.data_ptr()),
                                         C.c_void_p(D.data_ptr()) if D is not None else None,
                                         C.c_void_p(Db.data_ptr()) if Db is not None else None,
                                         C.c_void_p(pxy.data_ptr())))
        priors = {"mu_nu": sp.mu_nu, "sd_nu": sp.sd_nu}
        if sp.kind == "velocity":
            priors.update(mu_gamma=sp.mu_gamma, sd_gamma=sp.sd_gamma, mu_beta=sp.mu_beta, sd_beta=sp.sd_beta,
                          mu_nuw=sp.mu_nuw, sd_nuw=sp.sd_nuw)
        elif sp.with_delta_nu:
            sd = sp.sd_dnu
            sd = torch.full((sp.Nb, sp.Ng), float(sd)) if not torch.is_tensor(sd) else sd.reshape(sp.Nb, sp.Ng)
            priors["sd_dnu"] = sd
        for name, val in priors.items():
            t = _f32c(val).reshape(-1)
            self._check(lib.vc_set_prior(h, _lib.PRIORS.index(name), C.c_void_p(t.data_ptr()), t.numel()))
        for name, val in sp.condition_on.items():
            if name not in _lib.SITE_ID:
                raise ValueError(f"cannot condition on unknown site {name!r}")
            t = _f32c(val).reshape(sp.site_shape(name))
            if name == "ϕxy":
                t = t[sl]
            t = t.contiguous().reshape(-1)
            self._check(lib.vc_set_conditioned(h, _lib.SITE_ID[name], C.c_void_p(t.data_ptr()), t.numel()))
        self._check(lib.vc_finalize(h, self._stream()))
        self._counts_keepalive = None
        self.layout = _lib.vc_layout()
        self._check(lib.vc_get_layout(h, C.byref(self.layout)))
        L = self.layout
        self.header, self.n_global, self.n_local, self.total = L.header, L.n_global, L.n_local, L.total
        self.param_slices = {n: (L.offset[i], L.size[i]) for i, n in enumerate(_lib.PARAMS) if L.offset[i] >= 0}
        self.eps_slices = {n: (L.eps_offset[i], L.eps_size[i]) for i, n in enumerate(_lib.EPS) if L.eps_offset[i] >= 0}
        self.eps_total, self.eps_n_global = L.eps_total, L.eps_n_global
        self.params = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.loss_dev = torch.zeros(1, dtype=torch.float64, device=self.device)
        st = _lib.vc_stats()
        self._check(lib.vc_get_stats(h, C.byref(st)))
        self.stats = dict(algorithmic_bytes=st.algorithmic_bytes, streamed_bytes=st.streamed_bytes,
                          main_grid=st.main_grid, main_block=st.main_block, main_kind=st.main_kind,
                          main_kernel=st.main_kernel_name.decode(), hist_on_device=bool(st.hist_on_device),
                          setup_transient_bytes=st.setup_transient_bytes,
                          count_storage="u16" if st.count_storage_bytes == 2 else "f32",
                          pass_cells=[int(x) for x in st.pass_cells], launches_per_step=int(st.launches_per_step),
                          pw_inline=int(st.pw_inline), generic=bool(st.generic), onehot_batches=int(st.onehot_batches),
                          tail_spec=int(st.tail_spec), tail_spec_matched=int(st.tail_spec_matched), pw_lane=bool(st.pw_lane), hist_split=int(st.hist_split),
                          tail_spec_name=bytes(st.tail_spec_name).decode() or "generic")

    # ------------------------------------------------------------------------------------------
    def param_shape(self, name):
        sp = self.spec
        M = sp.Ng + sp.Nx * sp.Nhw
        return {"ν_locs": (sp.Ng, sp.Nh), "ν_scales": (sp.Ng, sp.Nh), "Δν_locs": (sp.Nb, sp.Ng),
                "νω_locs": (sp.Nx, sp.Nhw), "νω_scales": (sp.Nx, sp.Nhw), "loc": (M,),
                "cov_factor": (M, sp.rho_rank), "cov_diag": (M,),
                "ϕxy_locs": (self.Nc_local, 2)}.get(name, (sp.Ng,))

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, n = self.param_slices[name]
        return flat[off:off + n].view(self.param_shape(name))

    def named(self, flat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        flat = self.params if flat is None else flat
        return {n: self.view(flat, n) for n in self.param_slices}

    def init_params(self, cov_factor_draw: Optional[torch.Tensor] = None):
        """Initial `pyro.param` values of the guide, unconstrained (phase_inference_guide.py:36-45,
        velocity_inference_guide.py:25-43 / 78-102)."""
        sp = self.spec
        host = {}
        host["ν_locs"] = sp.mu_nu
        host["ν_scales"] = sp.sd_nu.log()
        if sp.with_delta_nu:
            host["Δν_locs"] = torch.full((sp.Nb, sp.Ng), float(sp.mu_dnu))
        if sp.kind == "velocity":
            host["logβg_locs"] = sp.mu_beta
            host["logβg_scales"] = sp.sd_beta.log()
            if sp.guide == "meanfield":
                host["logγg_locs"] = sp.mu_gamma
                host["logγg_scales"] = sp.sd_gamma.log()
                host["νω_locs"] = sp.mu_nuw
                host["νω_scales"] = sp.sd_nuw.log()
            else:
                if cov_factor_draw is None:
                    raise ValueError("LRMN guide: pass the first guide call's cov_factor draw")
                host["loc"] = torch.cat([sp.mu_gamma.reshape(-1), sp.mu_nuw.reshape(-1)])
                host["cov_factor"] = torch.clip(cov_factor_draw.float(), min=0).log()
                host["cov_diag"] = (torch.cat([sp.sd_gamma.reshape(-1), sp.sd_nuw.reshape(-1)]) ** 2).log()
                host["rho_real_loc"] = torch.full((sp.Ng,), float(sp.rho_mean))
        if sp.noisemodel == "NegativeBinomial":
            host["shape_inv_locs"] = torch.full((sp.Ng,), math.log(sp.gamma_alpha / sp.gamma_beta))
        host["ϕxy_locs"] = sp.phixy_prior[self.c0:self.c1]
        self.params.zero_()
        for n, v in host.items():
            self.view(self.params, n).copy_(_f32c(v).reshape(self.param_shape(n)))
        return self.params

    def set_params(self, named: Dict[str, torch.Tensor]):
        for n, v in named.items():
            v = _f32c(v)
            if n == "ϕxy_locs" and v.shape[0] == self.spec.Nc and self.Nc_local != self.spec.Nc:
                v = v[self.c0:self.c1]
            self.view(self.params, n).copy_(v.reshape(self.param_shape(n)))

    def pack_eps(self, eps: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Flat eps vector of this rank from the per-site draws of one guide call (global ϕxy is sliced)."""
        out = torch.zeros(self.eps_total, dtype=torch.float32)
        for n, (off, size) in self.eps_slices.items():
            v = torch.as_tensor(eps[n]).float()
            if n == "ϕxy":
                v = v.reshape(-1, 2)
                if v.shape[0] == self.spec.Nc:
                    v = v[self.c0:self.c1]
            out[off:off + size] = v.reshape(-1)
        return out.to(self.device)

    # ------------------------------------------------------------------------------------------
    def elbo_grad(self, eps: Optional[torch.Tensor] = None, seed: int = 0, step: int = 0,
                  step_dev: Optional[torch.Tensor] = None, params: Optional[torch.Tensor] = None,
                  grad: Optional[torch.Tensor] = None, loss_buf: Optional[torch.Tensor] = None):
        """Launches one ELBO + gradient evaluation on the current stream (asynchronous).
        Results: self.grad (header = loss hi/lo) and self.loss_dev (or slot step % len(loss_buf) of
        `loss_buf`, a float64 device ring).  `step_dev` (device int64[1]) is read as the step index and
        incremented by the call."""
        params = self.params if params is None else params
        grad = self.grad if grad is None else grad
        lb = self.loss_dev if loss_buf is None else loss_buf
        self._check(self.lib.vc_elbo_grad(
            self._h, C.c_void_p(params.data_ptr()),
            C.c_void_p(eps.data_ptr()) if eps is not None else None, C.c_uint64(seed), C.c_int64(step),
            C.c_void_p(step_dev.data_ptr()) if step_dev is not None else None,
            C.c_void_p(grad.data_ptr()), C.c_void_p(lb.data_ptr()), C.c_int64(lb.numel()), self._stream()))

    def svi_step(self, m, v, lr, lrd, b1, b2, adam_eps, clip, eps=None, seed=0, step=0, step_dev=None,
                 loss_buf=None):
        """ELBO + gradient + ClippedAdam in four launches (single rank); m, v: float32[total - header]."""
        lb = self.loss_dev if loss_buf is None else loss_buf
        self._check(self.lib.vc_svi_step(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_void_p(eps.data_ptr()) if eps is not None else None,
            C.c_uint64(seed), C.c_int64(step), C.c_void_p(step_dev.data_ptr()) if step_dev is not None else None,
            C.c_void_p(self.grad.data_ptr()), C.c_void_p(lb.data_ptr()), C.c_int64(lb.numel()),
            C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), lr, lrd, b1, b2, adam_eps, clip, self._stream()))

    def svi_step_fused(self, m, v, lr, lrd, b1, b2, adam_eps, clip, seed, step_dev, loss_buf=None, prime=False,
                       n_steps=1):
        """n_steps whole SVI steps, three launches each (vc_svi_run_fused): K_main -> K_tail -> K_omega, enqueued back to
        back from one call; `prime` draws the sample of the current step first (first call / after params were changed
        from outside)."""
        lb = self.loss_dev if loss_buf is None else loss_buf
        self._check(self.lib.vc_svi_run_fused(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_uint64(seed), C.c_void_p(step_dev.data_ptr()),
            C.c_void_p(self.grad.data_ptr()), C.c_void_p(lb.data_ptr()), C.c_int64(lb.numel()),
            C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), lr, lrd, b1, b2, adam_eps, clip, int(bool(prime)),
            C.c_int64(n_steps), self._stream()))

    def set_loss_every(self, k: int):
        """Opt-in (vc_set_loss_every): the fused single-rank runs form the loss at every k-th step only and run the gradient-only
        likelihood kernel in between (NB noise on the fast kernel set; the tutorial flow's velocity stage gains most).  k = 1 restores
        the default.  Raises HipEngineError (VC_ERR_UNSUPPORTED) when the configuration has no gradient-only kernel."""
        if int(k) == 1 and not hasattr(self.lib, "vc_set_loss_every"):
            return                       # (an older build of the ABI selected with VC_LIB_PATH + VC_LIB_OLDER=1: it has no period to reset)
        self._check(self.lib.vc_set_loss_every(self._h, C.c_int32(int(k))))

    def svi_run_particles(self, grad_acc, m, v, lr, lrd, b1, b2, adam_eps, clip, seed, step_dev, step0, num_particles, n_steps,
                          loss_buf=None):
        """n_steps steps of Trace_ELBO(num_particles=K) enqueued from one call (vc_svi_run_particles): per step K x the unfused
        kernel sequence on the Philox streams (seed, t K + k), the gradients averaged on the device, one ClippedAdam."""
        lb = self.loss_dev if loss_buf is None else loss_buf
        self._check(self.lib.vc_svi_run_particles(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_uint64(seed), C.c_void_p(step_dev.data_ptr()), C.c_int64(step0),
            C.c_void_p(self.grad.data_ptr()), C.c_void_p(grad_acc.data_ptr()), C.c_void_p(lb.data_ptr()), C.c_int64(lb.numel()),
            C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), lr, lrd, b1, b2, adam_eps, clip, int(num_particles),
            C.c_int64(n_steps), self._stream()))

    # ---- cells sharded over ranks: the fused step cut at its one exchange (vc_svi_run_sharded) ----------------------
    def exchange_size(self) -> int:
        n = C.c_int64()
        self._check(self.lib.vc_exchange_size(self._h, C.byref(n)))
        return int(n.value)

    def svi_run_sharded(self, xbuf, m, v, lr, lrd, b1, b2, adam_eps, clip, seed, step_dev, loss_buf=None, prime=False,
                        phase=_lib.VC_PHASE_AB, n_steps=1):
        """phase A: K_main + the part of the step before the exchange (writes `xbuf`, which the caller then sums over the
        ranks); phase B: the part after it; phase AB: n_steps whole steps, the sum made by the engine's own RCCL
        communicator (`init_rccl_comm`; a single rank needs none)."""
        lb = self.loss_dev if loss_buf is None else loss_buf
        self._check(self.lib.vc_svi_run_sharded(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_uint64(seed), C.c_void_p(step_dev.data_ptr()),
            C.c_void_p(self.grad.data_ptr()), C.c_void_p(xbuf.data_ptr()), C.c_void_p(lb.data_ptr()), C.c_int64(lb.numel()),
            C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()), lr, lrd, b1, b2, adam_eps, clip, int(bool(prime)), int(phase),
            C.c_int64(n_steps), self._stream()))

    @staticmethod
    def rccl_path() -> str:
        """The librccl.so this process already uses (PyTorch's bundled one), so that the engine's communicator and
        torch.distributed share one RCCL."""
        import os
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return cand if os.path.exists(cand) else "librccl.so"

    def init_rccl_comm(self, process_group=None) -> bool:
        """Creates the engine's own RCCL communicator over the ranks of `process_group` (collective).  The 128-byte unique
        id is made by group rank 0 through the library and broadcast with torch.distributed (any backend).  Returns True
        when EVERY rank succeeded (agreed by a MIN all-reduce), else False on every rank."""
        import torch.distributed as dist
        # idempotent: the engine is kept across fit() calls and every fit() builds a new SVIRunner; the communicator made
        # by the first one is the engine's for its lifetime.  (Every rank made it together, so every rank returns here together.)
        ranks = self._group_ranks(process_group)
        if getattr(self, "_rccl_ready", False):
            if ranks != self._rccl_group:
                raise HipEngineError(f"this engine's RCCL communicator was created for the ranks {self._rccl_group}; a run over "
                                     f"the group {ranks} needs an engine of its own (ADVICE r4: a communicator is not silently reused "
                                     "for another group)")
            return True
        path = self.rccl_path().encode()
        ok = 1
        ident = [None]
        # every stage is exception-safe and every rank takes part in both collectives below whatever happened to it locally:
        # a rank that fails alone must not leave the others waiting in a collective it never joins
        if self.rank == 0:
            try:
                buf = (C.c_char * 128)()
                if self.lib.vc_comm_rccl_unique_id(path, buf) == _lib.VC_OK:
                    ident = [bytes(buf.raw)]
            except Exception:
                ident = [None]
        src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
        dist.broadcast_object_list(ident, src=src, group=process_group)
        # a rank that cannot create the communicator must say so BEFORE the others enter ncclCommInitRank (which would wait
        # for it): first agree that every rank holds the id and the entry point ...
        have = int(ident[0] is not None and hasattr(self.lib, "vc_comm_init_rccl"))
        dev = self.device if dist.get_backend(process_group) == "nccl" else "cpu"
        flag = torch.tensor([have], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        if int(flag.item()) == 0:
            return False
        try:        # ... then create it (collective inside RCCL) ...
            ib = (C.c_char * 128).from_buffer_copy(ident[0])
            if self.lib.vc_comm_init_rccl(self._h, path, ib) != _lib.VC_OK:
                ok = 0
        except Exception:
            ok = 0
        # ... and agree on the outcome
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        self._rccl_ready = bool(int(flag.item()))
        self._rccl_group = ranks
        return self._rccl_ready

    @staticmethod
    def _group_ranks(process_group):
        """Global ranks of a torch.distributed group (the identity a communicator / IPC mapping was built for)."""
        import torch.distributed as dist
        try:
            return tuple(dist.get_process_group_ranks(process_group if process_group is not None else dist.group.WORLD))
        except Exception:
            return tuple(range(dist.get_world_size(process_group)))

    def comm_allreduce(self, buf: torch.Tensor):
        """Sum of a float32 device buffer over the ranks of the engine's own RCCL communicator, in place, on the current
        stream (vc_comm_allreduce) -- what the sharded run does between its two phases, as a call of its own."""
        self._check(self.lib.vc_comm_allreduce(self._h, C.c_void_p(buf.data_ptr()), C.c_int64(buf.numel()), self._stream()))

    def init_p2p_exchange(self, process_group=None) -> bool:
        """The one-shot peer-to-peer exchange (vc_p2p_alloc / vc_p2p_connect): this rank's region is created and exported,
        the 64-byte IPC handles of all ranks are gathered in rank order with torch.distributed (any backend), the peers'
        regions are mapped.  Returns True when EVERY rank is connected (MIN all-reduce), else False on every rank."""
        import torch.distributed as dist
        ranks = self._group_ranks(process_group)
        if getattr(self, "_p2p_ready", False):       # idempotent, like init_rccl_comm: one region per engine
            if ranks != self._p2p_group:
                raise HipEngineError(f"this engine's peer-to-peer region is mapped for the ranks {self._p2p_group}; a run over the "
                                     f"group {ranks} needs an engine of its own")
            return True
        buf = (C.c_char * 64)()
        rc = self.lib.vc_p2p_alloc(self._h, buf)
        mine = bytes(buf.raw) if rc == _lib.VC_OK else None
        world = dist.get_world_size(process_group)
        handles = [None] * world
        dist.all_gather_object(handles, mine, group=process_group)
        ok = 1
        if any(h is None for h in handles):
            ok = 0
        else:
            allh = (C.c_char * (64 * world)).from_buffer_copy(b"".join(handles))
            if self.lib.vc_p2p_connect(self._h, allh) != _lib.VC_OK:
                ok = 0
        flag = torch.tensor([ok], dtype=torch.int32,
                            device=self.device if dist.get_backend(process_group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        self._p2p_ready = bool(int(flag.item()))
        self._p2p_group = ranks
        return self._p2p_ready

    def clipped_adam(self, p, g, m, v, lr, lrd, b1, b2, eps, clip, t=0, t_dev=None, loss_hdr=None, loss_ring=None):
        """Fused HIP ClippedAdam on flat float32 buffers (same stream); optionally files the (all-reduced) loss
        found in `loss_hdr[0:2]` into `loss_ring[(t-1) % len]`."""
        rc = self.lib.vc_clipped_adam(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                      C.c_void_p(v.data_ptr()), p.numel(), lr, lrd, b1, b2, eps, clip, int(t),
                                      C.c_void_p(t_dev.data_ptr()) if t_dev is not None else None,
                                      C.c_void_p(loss_hdr.data_ptr()) if loss_hdr is not None else None,
                                      C.c_void_p(loss_ring.data_ptr()) if loss_ring is not None else None,
                                      loss_ring.numel() if loss_ring is not None else 0, self._stream())
        if rc != _lib.VC_OK:
            raise HipEngineError("vc_clipped_adam failed")

    def frozen_mask(self):
        """uint8 device tensor over the flat parameter buffer: 1 = a parameter tensor PyroOptim never steps because its site is
        conditioned (svi.frozen_param_names) -- what weight decay must leave alone; None when nothing is conditioned."""
        if getattr(self, "_frozen_mask", None) is None:
            from .svi import frozen_param_names
            names = frozen_param_names(self.spec) & set(self.param_slices)
            if not names:
                self._frozen_mask = False
            else:
                fm = torch.zeros(self.total, dtype=torch.uint8)
                for n in names:
                    off, size = self.param_slices[n]
                    fm[off:off + size] = 1
                self._frozen_mask = fm.to(self.device)
        return None if self._frozen_mask is False else self._frozen_mask

    def set_optimizer(self, kind: str = "clipped_adam", weight_decay: float = 0.0, frozen=None):
        """Which optimiser the step entry points of this engine apply (vc_set_optimizer): "clipped_adam" (pyro.optim.ClippedAdam, the
        default) or "adam" (pyro.optim.Adam = torch.optim.Adam); `frozen` = frozen_mask() when weight decay is on."""
        k = {"clipped_adam": _lib.VC_OPT_CLIPPED_ADAM, "adam": _lib.VC_OPT_ADAM}[kind]
        if not hasattr(self.lib, "vc_set_optimizer"):
            if k == _lib.VC_OPT_CLIPPED_ADAM and weight_decay == 0.0:
                return                   # (an older build of the ABI: ClippedAdam without weight decay is all it has)
            raise HipEngineError("this build of the library has no vc_set_optimizer")
        self._opt_frozen = frozen        # (the engine keeps the pointer)
        self._check(self.lib.vc_set_optimizer(self._h, int(k), float(weight_decay),
                                              C.c_void_p(frozen.data_ptr()) if frozen is not None else None))

    def adam_update(self, kind, p, g, m, v, lr, lrd, b1, b2, eps, clip, wd, frozen=None, t=0, t_dev=None, loss_hdr=None, loss_ring=None):
        """The optimiser as one launch on flat float32 buffers (vc_adam_update): kind "clipped_adam" | "adam"."""
        k = {"clipped_adam": _lib.VC_OPT_CLIPPED_ADAM, "adam": _lib.VC_OPT_ADAM}[kind]
        if not hasattr(self.lib, "vc_adam_update"):
            if k != _lib.VC_OPT_CLIPPED_ADAM or wd != 0.0:
                raise HipEngineError("this build of the library has no vc_adam_update")
            return self.clipped_adam(p, g, m, v, lr, lrd, b1, b2, eps, clip, t=t, t_dev=t_dev, loss_hdr=loss_hdr, loss_ring=loss_ring)
        rc = self.lib.vc_adam_update(int(k), C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                     C.c_void_p(v.data_ptr()), p.numel(), lr, lrd, b1, b2, eps,
                                     clip if math.isfinite(clip) else 3.0e38, wd,
                                     C.c_void_p(frozen.data_ptr()) if frozen is not None else None, int(t),
                                     C.c_void_p(t_dev.data_ptr()) if t_dev is not None else None,
                                     C.c_void_p(loss_hdr.data_ptr()) if loss_hdr is not None else None,
                                     C.c_void_p(loss_ring.data_ptr()) if loss_ring is not None else None,
                                     loss_ring.numel() if loss_ring is not None else 0, self._stream())
        if rc != _lib.VC_OK:
            raise HipEngineError("vc_adam_update failed")

    def sample_guide(self, eps: Optional[torch.Tensor] = None, seed: int = 0, step: int = 0):
        """One guide draw + deterministic sites (no likelihood); read the values with read_site()."""
        self._check(self.lib.vc_sample_guide(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_void_p(eps.data_ptr()) if eps is not None else None,
            C.c_uint64(seed), C.c_int64(step), self._stream()))

    def _site_id_shape(self, name: str):
        if name in _lib.SITE_ID:
            shape = (self.Nc_local, 2) if name == "ϕxy" else self.spec.site_shape(name)
            return _lib.SITE_ID[name], tuple(shape)
        if name == "ϕ":
            return _lib.DET_PHI, (self.Nc_local,)
        if name == "ω":
            return _lib.DET_OMEGA, (self.Nc_local,)
        if name == "eps":
            return _lib.DET_EPS, (self.eps_total,)
        raise KeyError(name)

    def sample_posterior(self, names, n_draws: int, seed: int = 0, step0: int = 0) -> Dict[str, torch.Tensor]:
        """n_draws guide draws + deterministic sites, batched on the device: {site: (n_draws, *site shape) DEVICE
        tensor}.  Draw i equals sample_guide(seed=seed, step=step0 + i)."""
        names = list(names)
        ids, outs = [], {}
        for n in names:
            sid, shape = self._site_id_shape(n)
            ids.append(sid)
            outs[n] = torch.empty((n_draws,) + shape, dtype=torch.float32, device=self.device)
        id_arr = (C.c_int * len(ids))(*ids)
        ptr_arr = (C.c_void_p * len(ids))(*[outs[n].data_ptr() for n in names])
        self._check(self.lib.vc_sample_posterior(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_uint64(seed), C.c_int64(step0), C.c_int64(n_draws),
            len(ids), id_arr, ptr_arr, self._stream()))
        return outs

    def expected_logs(self, nu, phi, cf_avg: float, dnu=None, omega=None, logbeta=None, gamma=None):
        """ElogS, ElogS2[, ElogU, ElogU2] of posterior_sampling as (Ng, Nc_local) DEVICE tensors (vc_expected_logs)."""
        dev, sp = self.device, self.spec
        f = lambda t, shape: None if t is None else torch.as_tensor(t, dtype=torch.float32).reshape(shape).to(dev).contiguous()
        nu, phi = f(nu, (sp.Ng, sp.Nh)), f(phi, (self.Nc_local,))
        dnu = f(dnu, (sp.Nb, sp.Ng)) if sp.with_delta_nu else None
        vel = sp.kind == "velocity"
        omega, logbeta, gamma = (f(omega, (self.Nc_local,)), f(logbeta, (sp.Ng,)), f(gamma, (sp.Ng,))) if vel else (None,) * 3
        outs = [torch.empty((sp.Ng, self.Nc_local), dtype=torch.float32, device=dev) for _ in range(4 if vel else 2)]
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self._check(self.lib.vc_expected_logs(
            self._h, p(nu), p(dnu), p(phi), p(omega), p(logbeta), p(gamma), C.c_float(cf_avg), p(outs[0]), p(outs[1]),
            p(outs[2]) if vel else None, p(outs[3]) if vel else None, self._stream()))
        self._keep = (nu, dnu, phi, omega, logbeta, gamma)       # inputs must outlive the asynchronous launch
        return outs

    def loss(self) -> float:
        return float(self.loss_dev.item())

    def read_site(self, name: str) -> torch.Tensor:
        sid, shape = self._site_id_shape(name)
        out = torch.empty(shape, dtype=torch.float32)
        self._check(self.lib.vc_read_site(self._h, sid, C.c_void_p(out.data_ptr()), out.numel(), self._stream()))
        return out

    def status(self):
        """(ok, first_bad_step, n_bad): the device-side failure latch (vc_get_status) -- ok is False once any step of
        this engine produced a NaN / Inf loss; synchronises the stream."""
        first, n = C.c_int64(-1), C.c_int64(0)
        rc = self.lib.vc_get_status(self._h, C.byref(first), C.byref(n), self._stream())
        if rc not in (_lib.VC_OK, _lib.VC_ERR_NONFINITE):
            self._check(rc)
        return rc == _lib.VC_OK, int(first.value), int(n.value)

    def clear_status(self):
        self._check(self.lib.vc_clear_status(self._h, self._stream()))

    def device_clock_mhz(self, window_us: float = 200.0) -> float:
        """Shader clock measured on the device right now (vc_device_clock_mhz)."""
        mhz = C.c_double()
        rc = self.lib.vc_device_clock_mhz(C.c_double(window_us), C.byref(mhz), self._stream())
        if rc != _lib.VC_OK:
            raise HipEngineError("vc_device_clock_mhz failed")
        return mhz.value

    def histogram(self):
        """(ptr int32[2*Ng+1], values float32[n], multiplicities float32[n]): the per-gene count histograms vc_finalize
        built (vc_get_histogram)."""
        import numpy as np
        n = C.c_int64()
        self._check(self.lib.vc_get_histogram(self._h, C.byref(n), None, None, None))
        ptr = np.zeros(2 * self.spec.Ng + 1, dtype=np.int32)
        val, cnt = np.zeros(n.value, dtype=np.float32), np.zeros(n.value, dtype=np.float32)
        self._check(self.lib.vc_get_histogram(self._h, C.byref(n), ptr.ctypes.data_as(C.c_void_p),
                                              val.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p)))
        return ptr, val, cnt

    def dump_dbg_times(self, path: str):
        """profiles/tools: the time stamps of a -DVC_DBG_TIMES build of the library (vc_dbg_dump_times)."""
        self._check(self.lib.vc_dbg_dump_times(self._h, path.encode()))

    def signature(self):
        """The size-independent signature of this configuration (vc_dbg_signature; csrc/vc_common.h VcSig) -- the key of the
        compiled specialisations of the small kernels (csrc/vc_tail_spec_rows.inc; profiles/tools/print_signature.py)."""
        out = (C.c_int32 * 27)()
        self._check(self.lib.vc_dbg_signature(self._h, out, 27))
        return [int(x) for x in out]

    def set_timing(self, enable: bool):
        self._check(self.lib.vc_set_timing(self._h, int(enable)))

    def get_timing(self):
        """(total ms spent in the likelihood kernel, number of launches) since set_timing(True)."""
        ms, n = C.c_double(), C.c_int64()
        self._check(self.lib.vc_get_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.vc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
