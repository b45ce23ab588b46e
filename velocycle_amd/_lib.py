"""ctypes binding of libvelocycle_hip.so (C ABI: include/velocycle_hip.h).

There is no fallback: if the HIP library is missing or does not load, importing this module's
`load()` raises -- the product path never routes through a CPU implementation.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VC_LIB_PATH selects another build of the same ABI (A/B measurements of kernel variants, profiles/tools/ab_libs.sh)
# without touching the in-tree product library; it must still be a HIP build of this ABI -- there is no fallback.
LIB_PATH = os.environ.get("VC_LIB_PATH") or os.path.join(HERE, "libvelocycle_hip.so")

VC_ABI_VERSION = 2
VC_OK = 0
VC_PHASE_A, VC_PHASE_B, VC_PHASE_AB = 1, 2, 3
VC_OPT_CLIPPED_ADAM, VC_OPT_ADAM = 0, 1
VC_ERR_ARG, VC_ERR_HIP, VC_ERR_UNSUPPORTED, VC_ERR_STATE, VC_ERR_NONFINITE = -1, -2, -3, -4, -5
MODEL = {"phase": 0, "velocity": 1}
GUIDE = {"meanfield": 0, "lrmn": 1}
NOISE = {"NegativeBinomial": 0, "Poisson": 1, "Lognormal": 2}

# sample sites, parameters and eps blocks, named as in the reference
SITES = ["ϕxy", "ν", "Δν", "shape_inv", "logγg", "logβg", "νω", "rho_real"]
SITE_ID = {n: i for i, n in enumerate(SITES)}
DET_PHI, DET_OMEGA, DET_EPS = 16, 17, 18
PARAMS = ["ν_locs", "ν_scales", "Δν_locs", "logγg_locs", "logγg_scales", "logβg_locs", "logβg_scales",
          "νω_locs", "νω_scales", "shape_inv_locs", "loc", "cov_factor", "cov_diag", "rho_real_loc",
          "ϕxy_locs"]
POSITIVE_PARAMS = {"ν_scales", "logγg_scales", "logβg_scales", "νω_scales", "shape_inv_locs",
                   "cov_factor", "cov_diag"}
EPS = ["logγg", "logβg", "ν", "νω", "eps_W", "eps_D", "ϕxy"]
PRIORS = ["mu_nu", "sd_nu", "mu_gamma", "sd_gamma", "mu_beta", "sd_beta", "mu_nuw", "sd_nuw", "sd_dnu"]
VC_P_COUNT, VC_E_COUNT = len(PARAMS), len(EPS)


class vc_config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("guide", C.c_int32),
                ("noise", C.c_int32), ("with_delta_nu", C.c_int32), ("n_harmonics", C.c_int32),
                ("n_harmonics_w", C.c_int32), ("Nb", C.c_int32), ("Nx", C.c_int32),
                ("lrmn_rank", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("Ng", C.c_int64), ("Nc_local", C.c_int64), ("Nc_global", C.c_int64),
                ("cell_offset", C.c_int64),
                ("gamma_alpha", C.c_float), ("gamma_beta", C.c_float),
                ("sigma_ln_s", C.c_float), ("sigma_ln_u", C.c_float),
                ("rho_mean", C.c_float), ("rho_std", C.c_float), ("rho_scale", C.c_float),
                ("reserved0", C.c_float)]


class vc_tuning(C.Structure):
    _fields_ = [("genes_per_lane", C.c_int32), ("blocks_per_cu", C.c_int32), ("cells_per_wave", C.c_int32),
                ("pass_min_cw", C.c_int32), ("n_pass_shares", C.c_int32), ("pass_shares", C.c_float * 4),
                ("tail_cells", C.c_int32), ("count_storage", C.c_int32), ("host_hist", C.c_int32),
                ("hist_dense", C.c_int32), ("pw_inline", C.c_int32), ("no_tail2", C.c_int32),
                ("no_tail_merged", C.c_int32), ("force_generic", C.c_int32), ("particles_layout", C.c_int32),
                ("dense_batches", C.c_int32), ("p2p_timeout_s", C.c_float), ("no_tail_spec", C.c_int32),
                ("no_pw_lane", C.c_int32), ("p2p_separate", C.c_int32), ("p2p_one_launch", C.c_int32), ("reserved", C.c_int32 * 3)]


class vc_layout(C.Structure):
    _fields_ = [("header", C.c_int64), ("n_global", C.c_int64), ("n_local", C.c_int64),
                ("total", C.c_int64), ("offset", C.c_int64 * VC_P_COUNT), ("size", C.c_int64 * VC_P_COUNT),
                ("eps_n_global", C.c_int64), ("eps_total", C.c_int64),
                ("eps_offset", C.c_int64 * VC_E_COUNT), ("eps_size", C.c_int64 * VC_E_COUNT)]


class vc_stats(C.Structure):
    _fields_ = [("algorithmic_bytes", C.c_int64), ("streamed_bytes", C.c_int64),
                ("main_grid", C.c_int64), ("main_block", C.c_int64), ("main_kind", C.c_int32),
                ("hist_on_device", C.c_int32), ("main_kernel_name", C.c_char * 96),
                ("setup_transient_bytes", C.c_int64), ("count_storage_bytes", C.c_int64),
                ("pass_cells", C.c_int32 * 4), ("launches_per_step", C.c_int32), ("pw_inline", C.c_int32),
                ("generic", C.c_int32), ("onehot_batches", C.c_int32), ("tail_spec", C.c_int32), ("tail_spec_matched", C.c_int32),
                ("tail_spec_name", C.c_char * 32), ("pw_lane", C.c_int32), ("hist_split", C.c_int32)]


EXPORTS = {
    "vc_abi_version": (C.c_int, []),
    "vc_create": (C.c_int, [C.POINTER(vc_config), C.POINTER(C.c_void_p)]),
    "vc_destroy": (None, [C.c_void_p]),
    "vc_last_error": (C.c_char_p, [C.c_void_p]),
    "vc_set_tuning": (C.c_int, [C.c_void_p, C.POINTER(vc_tuning)]),
    "vc_get_tuning": (C.c_int, [C.c_void_p, C.POINTER(vc_tuning)]),
    "vc_dbg_dump_times": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vc_dbg_signature": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "vc_set_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]),
    "vc_set_counts_csr": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "vc_get_histogram": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p]),
    "vc_set_cell_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vc_set_prior": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "vc_set_conditioned": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "vc_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vc_get_layout": (C.c_int, [C.c_void_p, C.POINTER(vc_layout)]),
    "vc_elbo_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vc_svi_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "vc_svi_step_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int, C.c_void_p]),
    "vc_svi_run_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                   C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_int, C.c_int64, C.c_void_p]),
    "vc_set_loss_every": (C.c_int, [C.c_void_p, C.c_int32]),
    "vc_exchange_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "vc_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vc_svi_run_particles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_int, C.c_int64, C.c_void_p]),
    "vc_svi_run_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                     C.c_double, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "vc_p2p_alloc": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vc_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vc_comm_rccl_unique_id": (C.c_int, [C.c_char_p, C.c_void_p]),
    "vc_comm_init_rccl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "vc_clipped_adam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                                  C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vc_set_optimizer": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "vc_adam_update": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_void_p]),
    "vc_sample_guide": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_void_p]),
    "vc_sample_posterior": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "vc_expected_logs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vc_read_site": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "vc_get_stats": (C.c_int, [C.c_void_p, C.POINTER(vc_stats)]),
    "vc_get_status": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "vc_clear_status": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vc_device_clock_mhz": (C.c_int, [C.c_double, C.POINTER(C.c_double), C.c_void_p]),
    "vc_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "vc_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Loads libvelocycle_hip.so (after torch, so that both share one HIP runtime). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C velocycle_amd/csrc` (or __graft_entry__.build()); "
            "velocycle_amd has no CPU fallback")
    import torch  # noqa: F401  (its bundled libamdhip64 must be the one already mapped)
    lib = C.CDLL(LIB_PATH)
    # The structs of this file are the ones of ABI version VC_ABI_VERSION: a library of ANY other version is refused before a
    # single struct crosses (vc_get_stats writes the whole struct of ITS version).  VC_LIB_OLDER=1 only tolerates missing entry
    # points of an older build of the SAME version.
    lib.vc_abi_version.restype = C.c_int
    lib.vc_abi_version.argtypes = []
    got = lib.vc_abi_version()
    if got != VC_ABI_VERSION:
        raise HipLibraryError(f"{LIB_PATH}: ABI version {got}, this host side was written against {VC_ABI_VERSION} "
                              "(rebuild with `make -C velocycle_amd/csrc`)")
    older = bool(os.environ.get("VC_LIB_PATH")) and os.environ.get("VC_LIB_OLDER") == "1"
    for name, (res, args) in EXPORTS.items():
        if older and not hasattr(lib, name):
            continue                     # measurement aid: an OLDER build of the library selected with VC_LIB_PATH for a same-box A/B
        fn = getattr(lib, name)          # AttributeError if a symbol of the header is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
