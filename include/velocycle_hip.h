/*
 * velocycle_hip.h -- C ABI of the MI355X (gfx950) engine for VeloCycle's SVI hot path.
 *
 * One engine = one fit (one model + guide + data shard) on one GPU.  One process per GPU; the
 * caller (velocycle_amd/engine.py through ctypes) owns the flat parameter / gradient buffers
 * (PyTorch-ROCm tensors passed as device pointers) and does the optimiser update and the
 * all-reduce; the library owns the count matrices (re-laid-out in HBM), per-step workspaces and
 * all kernels.  No C++ exception crosses this boundary: every call returns 0 (VC_OK) or a negative
 * code, with a message available from vc_last_error().
 *
 * Reference interfaces each entry point replaces (reference = lamanno-epfl/velocycle v0.1.0.5; the
 * reference has no FFI of its own -- its "interface" for this path is the set of Python objects
 * that `pyro.infer.SVI.step` drives):
 *
 *   vc_create / vc_set_*        the MetaparContainer built by preprocess_for_phase_estimation
 *                               (velocycle/preprocessing.py:168-205) and
 *                               preprocess_for_velocity_estimation (preprocessing.py:270-323),
 *                               plus poutine.condition/block wiring of
 *                               PhaseFitModel.__init__ (phase_inference_model.py:109-123) and
 *                               VelocityFitModel.__init__ (velocity_inference_model.py:60-74)
 *   vc_get_layout               the pyro.param store created by the guides
 *                               (phase_inference_guide.py:36-45, velocity_inference_guide.py:25-43, 78-102)
 *   vc_elbo_grad                one Trace_ELBO(num_particles=1).loss_and_grads(model, guide, mp):
 *                               guide trace + model replay + log-prob sums + backward, i.e. the body
 *                               of `svi.step` (phase_inference_model.py:169, velocity_inference_model.py:120)
 *                               for phase_latent_variable_model/guide (phase_inference_model.py:343-395,
 *                               phase_inference_guide.py:10-56) and velocity_latent_variable_model/guide
 *                               [_LRMN] (velocity_inference_model.py:304-471, velocity_inference_guide.py:9-141)
 *   vc_sample_posterior         Predictive(model, guide=guide, num_samples=n) as used by posterior_sampling
 *                               (velocity_inference_model.py:189-291, phase_inference_model.py:203-302)
 *   vc_expected_logs            the ElogS / ElogU / ElogS2 / ElogU2 einsums of posterior_sampling
 *                               (velocity_inference_model.py:236-258, phase_inference_model.py:248-262)
 *   vc_read_site                the sampled / deterministic sites a trace exposes
 *                               (pyro.deterministic calls at velocity_inference_model.py:327-369)
 */
#ifndef VELOCYCLE_HIP_H
#define VELOCYCLE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): vc_stats ends with onehot_batches, tail_spec, tail_spec_matched, tail_spec_name[32], pw_lane, hist_split (a consumer built against
 * version 1 would have its smaller vc_stats overrun by vc_get_stats: vc_create refuses it); vc_tuning, vc_set_tuning / vc_get_tuning,
 * vc_dbg_signature, vc_set_optimizer / vc_adam_update exist. */
#define VC_ABI_VERSION 2

/* return codes */
#define VC_OK 0
#define VC_ERR_ARG (-1)          /* bad argument / shape */
#define VC_ERR_HIP (-2)          /* a HIP runtime call failed */
#define VC_ERR_UNSUPPORTED (-3)  /* configuration outside the compiled kernel set */
#define VC_ERR_STATE (-4)        /* call order violated (e.g. step before finalize) */
#define VC_ERR_NONFINITE (-5)    /* a step produced a NaN / Inf loss (vc_get_status) */

/* model / guide / noise selectors */
#define VC_MODEL_PHASE 0         /* phase_latent_variable_model */
#define VC_MODEL_VELOCITY 1      /* velocity_latent_variable_model[_LRMN] */
#define VC_GUIDE_MEANFIELD 0
#define VC_GUIDE_LRMN 1
#define VC_NOISE_NB 0            /* "NegativeBinomial" (GammaPoisson) */
#define VC_NOISE_POISSON 1
#define VC_NOISE_LOGNORMAL 2

/* sample sites (bit mask for conditioning, id for vc_set_conditioned / vc_read_site) */
#define VC_SITE_PHIXY 0      /* "ϕxy"       (Nc_local, 2) */
#define VC_SITE_NU 1         /* "ν"         (Ng, Nh)      */
#define VC_SITE_DNU 2        /* "Δν"        (Nb, Ng)      */
#define VC_SITE_SHAPE_INV 3  /* "shape_inv" (Ng)          */
#define VC_SITE_LOGGAMMA 4   /* "logγg"     (Ng)          */
#define VC_SITE_LOGBETA 5    /* "logβg"     (Ng)          */
#define VC_SITE_NUOMEGA 6    /* "νω"        (Nx, Nhw)     */
#define VC_SITE_RHO_REAL 7   /* "rho_real"  (Ng)          */
#define VC_SITE_COUNT 8
/* deterministic sites readable through vc_read_site */
#define VC_DET_PHI 16        /* "ϕ"     (Nc_local)        */
#define VC_DET_OMEGA 17      /* "ω"     (Nc_local)        */
#define VC_DET_EPS 18        /* the eps vector used by the last step (layout: vc_layout.eps_*) */

/* variational parameters (the guide's pyro.param store), all stored UNCONSTRAINED:
 * positive parameters are stored as their log (Pyro's transform_to(positive) = exp). */
#define VC_P_NU_LOCS 0          /* "ν_locs"        (Ng, Nh) */
#define VC_P_NU_USCALES 1       /* "ν_scales"      (Ng, Nh)  log */
#define VC_P_DNU_LOCS 2         /* "Δν_locs"       (Nb, Ng) */
#define VC_P_LOGGAMMA_LOCS 3    /* "logγg_locs"    (Ng) */
#define VC_P_LOGGAMMA_USCALES 4 /* "logγg_scales"  (Ng)      log */
#define VC_P_LOGBETA_LOCS 5     /* "logβg_locs"    (Ng) */
#define VC_P_LOGBETA_USCALES 6  /* "logβg_scales"  (Ng)      log */
#define VC_P_NUOMEGA_LOCS 7     /* "νω_locs"       (Nx, Nhw) */
#define VC_P_NUOMEGA_USCALES 8  /* "νω_scales"     (Nx, Nhw) log */
#define VC_P_SHAPE_INV_ULOCS 9  /* "shape_inv_locs"(Ng)      log */
#define VC_P_LRMN_LOC 10        /* "loc"           (Ng + Nx*Nhw) */
#define VC_P_LRMN_UCOV_FACTOR 11/* "cov_factor"    (Ng + Nx*Nhw, rank) log */
#define VC_P_LRMN_UCOV_DIAG 12  /* "cov_diag"      (Ng + Nx*Nhw) log */
#define VC_P_RHO_REAL_LOC 13    /* "rho_real_loc"  (Ng) */
#define VC_P_PHIXY_LOCS 14      /* "ϕxy_locs"      (Nc_local, 2)   -- the only rank-local block */
#define VC_P_COUNT 15

/* standard-normal draws of one guide call */
#define VC_E_LOGGAMMA 0   /* (Ng)       mean-field only */
#define VC_E_LOGBETA 1    /* (Ng)       */
#define VC_E_NU 2         /* (Ng, Nh)   */
#define VC_E_NUOMEGA 3    /* (Nx, Nhw)  mean-field only */
#define VC_E_LRMN_W 4     /* (rank)     LRMN only */
#define VC_E_LRMN_D 5     /* (Ng + Nx*Nhw) LRMN only */
#define VC_E_PHIXY 6      /* (Nc_local, 2)  -- rank-local block */
#define VC_E_COUNT 7

/* per-gene / global prior arrays for vc_set_prior (host pointers, float32) */
#define VC_PRIOR_MU_NU 0      /* μνg (Ng, Nh) */
#define VC_PRIOR_SD_NU 1      /* σνg (Ng, Nh) */
#define VC_PRIOR_MU_GAMMA 2   /* μγ (Ng) */
#define VC_PRIOR_SD_GAMMA 3
#define VC_PRIOR_MU_BETA 4
#define VC_PRIOR_SD_BETA 5
#define VC_PRIOR_MU_NUOMEGA 6 /* μνω (Nx, Nhw) */
#define VC_PRIOR_SD_NUOMEGA 7
#define VC_PRIOR_SD_DNU 8     /* σΔν (Nb, Ng) -- phase model; velocity hard-codes 0.01 (velocity_inference_model.py:332) */
#define VC_PRIOR_COUNT 9

typedef struct vc_engine vc_engine;

typedef struct vc_config {
  int32_t abi_version;     /* = VC_ABI_VERSION */
  int32_t model;           /* VC_MODEL_* */
  int32_t guide;           /* VC_GUIDE_* (phase: MEANFIELD) */
  int32_t noise;           /* VC_NOISE_* */
  int32_t with_delta_nu;   /* 0/1 */
  int32_t n_harmonics;     /* H   (Nh  = 2H+1), 1..3 */
  int32_t n_harmonics_w;   /* Hw  (Nhw = 2Hw+1), velocity only */
  int32_t Nb;              /* batches (rows of Db) */
  int32_t Nx;              /* conditions (rows of D), velocity only */
  int32_t lrmn_rank;       /* rho_rank (5) */
  int32_t rank;            /* data-parallel rank; rank 0 adds the replicated (gene-level) prior/entropy terms */
  int32_t world_size;
  int64_t Ng;              /* genes */
  int64_t Nc_local;        /* cells held by this rank */
  int64_t Nc_global;       /* cells over all ranks */
  int64_t cell_offset;     /* global index of this rank's first cell (eps stream slicing) */
  float gamma_alpha, gamma_beta;      /* shape_inv ~ Gamma(alpha, beta) */
  float sigma_ln_s, sigma_ln_u;       /* Lognormal noise scales (phase 0.5; velocity 0.1, 0.1) */
  float rho_mean, rho_std, rho_scale; /* LRMN */
  float reserved0;
} vc_config;

/* Tuning of the engine: DATA handed over by the caller, never ambient state (the library reads no environment variable).
 * All-zero = the measured defaults of DESIGN.md.  The reference has no counterpart (its "tuning" is the kwargs of
 * preprocess_for_* and the dict given to ClippedAdam: preprocessing.py:103-122, 207-240); SURVEY.md section 5 asks for "engine
 * config = plain C struct mirrored by a Python dataclass": this is it (velocycle_amd.tuning.Tuning).  Every rank of a sharded
 * run must pass the same values (the exchange buffer's layout follows genes_per_lane); the host side checks that. */
typedef struct vc_tuning {
  int32_t genes_per_lane;    /* 0: the rule of vc_finalize | 4 | 8 (8 is honoured only where that kernel has no scratch) */
  int32_t blocks_per_cu;     /* 0: the occupancy the code object reports | n: tile the likelihood kernel for n resident workgroups per CU */
  int32_t cells_per_wave;    /* 0: one balanced resident round | n: fixed cells per wave (ragged tiles in tests; no pass shares) */
  int32_t pass_min_cw;       /* cells per wave below which the dispatch passes take equal shares; 0 = 12 */
  int32_t n_pass_shares;     /* 0: measured defaults | 1: equal shares | 2..4: pass_shares[0..n) (later passes continue the last ratio) */
  float pass_shares[4];
  int32_t tail_cells;        /* cells per cell block of the second launch: 0 auto | 256 | 512 | 1024 */
  int32_t count_storage;     /* 0: uint16 whenever every count is an integer <= 65535 | 1: keep float32 */
  int32_t host_hist;         /* 1: per-gene count histograms by the host pass (the checker of the device pass) */
  int32_t hist_dense;        /* histogram form of the negative binomial's lgamma terms: 0 auto | 1 (value, multiplicity) lists | 2 dense tables */
  int32_t pw_inline;         /* the likelihood kernel's own d loglik / d nu_omega partials: 0 auto | 1 never | 2 even where they cost a resident workgroup */
  int32_t no_tail2;          /* 1: three launches per single-rank step where two (vc_tail2_kernel) are the default */
  int32_t no_tail_merged;    /* 1: the tutorial flow without its merged second launch */
  int32_t force_generic;     /* 1: the run-time-sized kernel set on a configuration the compiled fast set covers */
  int32_t particles_layout;  /* vc_svi_run_particles: 0 batched (K + 3 launches) | 1 serial | 2 streams */
  int32_t dense_batches;     /* 1: batch offsets as the dense Db contraction even when Db is one-hot (A/B, tests) */
  float p2p_timeout_s;       /* bound of the peer-to-peer exchange's wait; 0 = 2 s */
  int32_t no_tail_spec;      /* 1: the run-time-flag small kernels even where an instantiation compiled for this configuration exists (A/B, tests) */
  int32_t no_pw_lane;        /* 1: the U-only kernel reduces its per-cell sums over the wave even where the per-lane accumulation of the
                                nu_omega partials applies (one condition, D == 1; A/B, tests) */
  int32_t p2p_separate;      /* 1: the peer-to-peer exchange as a launch of its own between phases A and B (rounds 3-5); 0: folded into
                                phase B (its blocks pass the exchange's gate and add the ranks' slots where they read them) */
  int32_t p2p_one_launch;    /* 1 (with the peer-to-peer exchange): phases A and B of a sharded step in ONE launch, the exchange at block
                                granularity inside it -- K_main + one launch per step; needs 256-cell cell blocks and gene + cell + loss
                                blocks <= 240 (else the folded three-launch step).  Opt-in: it has run between processes on one device only */
  int32_t reserved[3];
} vc_tuning;

typedef struct vc_layout {
  int64_t header;                 /* floats reserved at the front of params/grad (grad[0..1] = loss hi/lo) */
  int64_t n_global;               /* floats of replicated parameters (after the header) */
  int64_t n_local;                /* floats of rank-local parameters (ϕxy_locs) */
  int64_t total;                  /* header + n_global + n_local */
  int64_t offset[VC_P_COUNT];     /* absolute float offset of each parameter block, -1 if absent */
  int64_t size[VC_P_COUNT];
  int64_t eps_n_global, eps_total;
  int64_t eps_offset[VC_E_COUNT]; /* float offset in the eps vector, -1 if absent */
  int64_t eps_size[VC_E_COUNT];
} vc_layout;

typedef struct vc_stats {
  int64_t algorithmic_bytes;      /* count-matrix bytes one step must read in the reference's fp32 storage */
  int64_t streamed_bytes;         /* bytes the main kernel actually streams (padded layout, uint16 or float32 elements) */
  int64_t main_grid, main_block;  /* launch geometry of the likelihood kernel */
  int32_t main_kind;              /* 0 phase(S) 1 velocity(S+U) 2 velocity, S-term hoisted (U only) */
  int32_t hist_on_device;         /* 1: the per-gene count histograms were built on the device during the re-layout */
  char main_kernel_name[96];
  int64_t setup_transient_bytes;  /* device memory vc_finalize held only while it ran (histogram tables, owned uploads) */
  int64_t count_storage_bytes;    /* bytes per count element in HBM: 2 (uint16: every count an integer <= 65535) or 4 */
  int32_t pass_cells[4];          /* cells per wave of the likelihood kernel's workgroups in dispatch pass 0..3 (all equal:
                                     balanced tiling; falling: the older passes take larger shares) */
  int32_t launches_per_step;      /* kernel launches of one steady-state step of vc_svi_run_fused on this engine: 2 or 3 */
  int32_t pw_inline;              /* floats per row of the likelihood kernel's own d loglik / d nu_omega partials (4 | 8), 0: off */
  int32_t generic;                /* 1: the run-time-sized (slower) kernel set is in use: a configuration outside the compiled
                                     fast set (H > 3, > 4 batches, LRMN rank > 8, > 64 angular-speed coefficients) */
  int32_t onehot_batches;         /* n > 0: the batch design matrix Db is one-hot, its n batch offsets are folded into the constant
                                     harmonic per workgroup of the likelihood kernel (the kernel's NB is 0: nothing per cell) */
  int32_t tail_spec;              /* > 0: the small kernels of this engine's fused steps (vc_svi_run_fused on one rank, vc_svi_run_sharded
                                     on a shard) RUN in the instantiation compiled for this configuration (row of
                                     csrc/vc_tail_spec_rows.inc, name in tail_spec_name); 0: run-time flags */
  int32_t tail_spec_matched;      /* the row the configuration's signature matched, whether or not that row is compiled for the launch
                                     structure in use (e.g. a "*_rank" row matched by a single-rank engine kept at three launches) */
  char tail_spec_name[32];
  int32_t pw_lane;                /* 1: the U-only likelihood kernel accumulates its d loglik / d nu_omega partials per lane from the cell
                                     record (one condition, D == 1: W_c = (1, sin k phi_c, cos k phi_c)) and stores no per-cell rows */
  int32_t hist_split;             /* gene blocks (per count matrix) whose dense histogram sums the one-launch tail evaluates in four quarter blocks:
                                     those whose largest count exceeds 255; 0 with the (value, multiplicity) lists (was reserved: same layout) */
} vc_stats;

/* lifecycle ------------------------------------------------------------------------------- */
int vc_abi_version(void);
int vc_create(const vc_config* cfg, vc_engine** out);
void vc_destroy(vc_engine* e);
/* message of the last failing call on `e` (or of the last failing vc_create when e == NULL) */
const char* vc_last_error(const vc_engine* e);

/* Tuning (optional): after vc_create and before the first vc_set_counts* call; NULL or never calling it = all defaults.
 * VC_ERR_ARG for a value outside the ranges documented at vc_tuning, VC_ERR_STATE once counts have been handed over. */
int vc_set_tuning(vc_engine* e, const vc_tuning* t);
/* The tuning in effect (what vc_set_tuning stored; all-zero if it was never called). */
int vc_get_tuning(const vc_engine* e, vc_tuning* out);
/* Builds of the library with -DVC_DBG_TIMES only (profiles/tools): writes the per-wave / per-block time stamps of the last
 * launches to `path`; VC_ERR_UNSUPPORTED in the product build. */
int vc_dbg_dump_times(vc_engine* e, const char* path);
/* the size-independent signature of the finalized configuration (csrc/vc_common.h: VcSig), n >= 27 ints: what
 * profiles/tools/print_signature.py turns into a row of csrc/vc_tail_spec_rows.inc */
int vc_dbg_signature(const vc_engine* e, int32_t* out, int n);

/* inputs (call before vc_finalize) ---------------------------------------------------------- */
/* Count matrices, element (g, c) at ptr[g*gene_stride + c*cell_stride] (so both the reference's
 * `S.T.float()` view -- gene_stride 1, cell_stride Ng -- and a contiguous (Ng,Nc) array work).
 * U may be NULL for the phase model.  on_device: pointers are device (1) or host (0) memory.  Device memory is NOT
 * copied: it is read by vc_finalize and must stay valid until vc_finalize has returned.  Host memory is uploaded
 * row by row (this rank's cells only).  Values must be finite and >= 0 (checked on the device; vc_finalize returns
 * VC_ERR_ARG otherwise). */
int vc_set_counts(vc_engine* e, const float* S, const float* U, int64_t gene_stride,
                  int64_t cell_stride, int on_device);
/* The same matrices as canonical CSR (rows = this rank's cells, columns = genes; no duplicate entries): what AnnData
 * layers hold before the reference densifies them (preprocessing.py:141-147 `.A`, 243-252).  which: 0 = spliced,
 * 1 = unspliced (velocity model).  indptr int64[Nc_local + 1], indices int32[nnz], data float[nnz]; on_device as above
 * (device arrays must stay valid until vc_finalize has returned).  The dense matrix is never formed on the host: the
 * blocked HBM layout is filled by a scatter kernel that also builds the per-gene count histograms. */
int vc_set_counts_csr(vc_engine* e, int which, const int64_t* indptr, const int32_t* indices, const float* data,
                      int64_t nnz, int on_device);
/* count_factor (Nc_local), D (Nx, Nc_local) row-major [NULL for phase], Db (Nb, Nc_local) row-major
 * [NULL when Nb == 0], phixy_prior (Nc_local, 2).  Host pointers. */
int vc_set_cell_data(vc_engine* e, const float* count_factor, const float* D, const float* Db,
                     const float* phixy_prior);
int vc_set_prior(vc_engine* e, int which, const float* data, int64_t n);      /* host pointer */
/* poutine.condition: fix a sample site to `values` (host pointer, canonical shape of VC_SITE_*). */
int vc_set_conditioned(vc_engine* e, int site, const float* values, int64_t n);
/* Builds the HBM layout, per-gene count histograms, hoisted constants and workspaces. */
int vc_finalize(vc_engine* e, void* hip_stream);

/* hot path ---------------------------------------------------------------------------------- */
int vc_get_layout(const vc_engine* e, vc_layout* out);
/* One ELBO + gradient evaluation.  params/grad: device float[layout.total]; eps: device
 * float[layout.eps_total] or NULL (then eps = Philox4x32-10(seed, step, index), identical on every
 * rank for replicated sites and sliced by cell_offset for ϕxy).  step_dev: optional device int64
 * read instead of `step` and INCREMENTED by one at the end of the call (lets the call be replayed
 * from a captured hipGraph).  loss_dev: device double[loss_slots] (loss_slots >= 1); this rank's loss
 * contribution is written to slot step % loss_slots (and as float hi/lo to grad[0..1]).
 * Asynchronous on `hip_stream`; no allocation, no synchronisation. */
int vc_elbo_grad(vc_engine* e, const float* params, const float* eps, uint64_t seed, int64_t step,
                 int64_t* step_dev, float* grad, double* loss_dev, int64_t loss_slots, void* hip_stream);

/* pyro.optim.ClippedAdam (pyro-ppl 1.8.6 optim/clipped_adam.py; call sites: tutorial cells 27/43/56)
 * as ONE launch on a flat buffer of n floats: lr_t = lr*lrd^t, g = clamp(g, +-clip_norm),
 * m/v moments, p -= lr_t*sqrt(1-beta2^t)/(1-beta1^t) * m/(sqrt(v)+eps).  t is the 1-based step, read
 * from t_dev (device int64) when non-NULL.  An alternative to the PyTorch-op update of svi.py.
 * loss_hdr / loss_ring (optional, may be NULL): after an all-reduce of the gradient header the summed loss
 * (float hi + lo at loss_hdr[0..1]) is stored as a double into loss_ring[(t-1) % loss_slots]. */
int vc_clipped_adam(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    double lr, double lrd, double beta1, double beta2, double eps, double clip_norm,
                    int64_t t, const int64_t* t_dev, const float* loss_hdr, double* loss_ring,
                    int64_t loss_slots, void* hip_stream);

/* The optimisers fit() accepts.  The reference hands whatever PyroOptim object it is given to pyro.infer.SVI
 * (velocity_inference_model.py:76-84,111; phase_inference_model.py:125-133,162): every package tutorial passes
 * pyro.optim.ClippedAdam (cells 27/43/56), tutorials/1D_Pancreas_Analysis.ipynb cell 26 passes pyro.optim.Adam (= torch.optim.Adam).
 *   VC_OPT_CLIPPED_ADAM  lr_t = lr * lrd^t, g = clamp(g, +-clip_norm), [g += weight_decay * p], m / v moments,
 *                        p -= lr_t sqrt(1 - beta2^t) / (1 - beta1^t) * m / (sqrt(v) + eps)
 *   VC_OPT_ADAM          [g += weight_decay * p], no clamp, no decay (lrd and clip_norm are ignored),
 *                        p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)       (amsgrad / maximize: not supported)
 * vc_set_optimizer selects what the step entry points of an engine apply (default: ClippedAdam, no weight decay);
 * vc_adam_update is the optimiser as a call of its own (vc_clipped_adam = kind VC_OPT_CLIPPED_ADAM, weight_decay 0).
 * `frozen` (DEVICE bytes, one per float of the flat parameter buffer / of the n updated floats; may be NULL; caller-owned, must stay
 * valid while the engine uses it): 1 marks a parameter TENSOR that has no path to the loss because its sample site is conditioned
 * (poutine.block: velocity_inference_model.py:65-66) -- PyroOptim never steps such a tensor (its .grad is None), so weight decay
 * must not move it either; without weight decay a frozen tensor stays where it is by itself (zero gradient, zero moments). */
#define VC_OPT_CLIPPED_ADAM 0
#define VC_OPT_ADAM 1
int vc_set_optimizer(vc_engine* e, int kind, double weight_decay, const uint8_t* frozen);
int vc_adam_update(int kind, float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double lrd,
                   double beta1, double beta2, double eps, double clip_norm, double weight_decay, const uint8_t* frozen, int64_t t,
                   const int64_t* t_dev, const float* loss_hdr, double* loss_ring, int64_t loss_slots, void* hip_stream);

/* One whole SVI step (single rank): vc_elbo_grad with pyro's ClippedAdam merged into its last kernel -- 4
 * launches instead of 5.  exp_avg / exp_avg_sq: device float[total - header], zero-initialised by the caller;
 * the optimiser step is step + 1.  Equivalent to vc_elbo_grad followed by vc_clipped_adam on params[header:].
 * grad still receives the gradient and the loss header.  Returns VC_ERR_STATE when world_size > 1: with sharded
 * cells the all-reduce has to sit between the two halves. */
int vc_svi_step(vc_engine* e, float* params, const float* eps, uint64_t seed, int64_t step, int64_t* step_dev,
                float* grad, double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr,
                double lrd, double beta1, double beta2, double adam_eps, double clip_norm, void* hip_stream);

/* One whole SVI step (single rank) in THREE launches, every O(Ng + Nc) latency chain of the step run once: the
 * likelihood kernel; a per-gene-block / per-cell-block kernel that finishes the reductions, applies the chain rule,
 * pyro's ClippedAdam AND draws the guide sample of the next step from the fresh parameters (gene table, cell table,
 * prior / guide log-densities); and a small kernel for the one global dependency, the angular-speed coefficients
 * (gradient, optimiser, next sample, omega_c into the cell records) that also assembles the loss of the finished step.
 * eps is the Philox stream (seed, step, index) only.  step_dev (device int64, required) holds the number of finished
 * steps t: the call evaluates step t, stores its loss into loss_dev[t % loss_slots] and leaves step_dev = t + 1.
 * prime != 0: first draw the sample of step t from `params` as they are (needed for the first call, and again after the
 * caller has changed params / step_dev / seed behind the engine's back, e.g. on resume); prime = 0 continues from the
 * tables the previous call left.  Same trajectory as vc_svi_step (tests/test_hip_fused.py).  VC_ERR_STATE when
 * world_size > 1. */
int vc_svi_step_fused(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, float* grad, double* loss_dev,
                      int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr, double lrd, double beta1,
                      double beta2, double adam_eps, double clip_norm, int prime, void* hip_stream);
/* n_steps of the above, back to back (the body of `for step in range(num_steps): svi.step(...)`,
 * velocity_inference_model.py:118-121, with verbose=False): 3 * n_steps asynchronous launches on hip_stream from one
 * call, no host round trip in between -- measured faster than replaying the same launches from a hipGraph (ROCm 7.2:
 * ~2 us per kernel node, profiles/r02_step_overhead.md).  loss_slots >= n_steps keeps every loss of the run. */
int vc_svi_run_fused(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, float* grad, double* loss_dev,
                     int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr, double lrd, double beta1,
                     double beta2, double adam_eps, double clip_norm, int prime, int64_t n_steps, void* hip_stream);

/* --- opt-in: the loss every k-th step only ("or every k steps in perf mode") -----------------------------
 * The reference's loop takes the loss of EVERY step from `svi.step` (velocity_inference_model.py:118-121); that stays the default
 * (k = 1).  vc_set_loss_every(e, k > 1) lets the launches of vc_svi_run_fused run a gradient-only instantiation of the likelihood
 * kernel except at every k-th one (counted from this call: launches 0, k, 2k, ... form the loss).  What that instantiation leaves
 * out is what serves the loss VALUE only: with phi_xy, nu, delta nu and shape_inv conditioned (the tutorials' velocity stage)
 * both logarithms per element (the gradients then agree with the full kernel's to float32 rounding: mu = 2^eta * z instead of
 * 2^(eta + log2 z)); with shape_inv learned only log2(z) and the loss accumulations (the same gradient bits).  The loss slots of
 * the other steps hold the prior / guide terms without the likelihood -- the caller must not read them as losses (velocycle_amd
 * reports NaN there).  Negative-binomial noise on the compiled fast kernel set; VC_ERR_UNSUPPORTED otherwise.
 * vc_svi_step_fused, vc_svi_run_sharded, vc_elbo_grad and vc_svi_run_particles always form the loss. */
int vc_set_loss_every(vc_engine* e, int32_t k);

/* --- Trace_ELBO(num_particles = K) from one call (single rank) -----------------------------------------------------------
 * Replaces, for `fit(loss=Trace_ELBO(num_particles=K))` (velocity_inference_model.py:79,111, phase_inference_model.py:128,162:
 * the user's ELBO object goes into SVI; K guide draws per step, loss and gradients averaged before the optimiser step), the
 * host loop  K x vc_elbo_grad + average + vc_clipped_adam:  n_steps steps are enqueued from this one call, per step
 *   particle k < K:  K_pre (Philox stream (seed, t K + k), t read from step_dev) -> K_main -> K_post -> K_fin, with its own
 *               per-step workspaces and gradient buffer and, for k >= 1, on a HIP stream of the engine's (created at the first
 *               call; the parameters do not change between particles, so every K_pre starts at once, the likelihood kernels
 *               follow one another, K_post / K_fin of particle k run beside K_main of particle k + 1);
 *   hip_stream joins them: the K gradients and losses are added in particle order and multiplied by 1 / K -- left in `grad`
 *   (header: the averaged loss, hi / lo) and in loss_dev[t % loss_slots]; step_dev := t + 1; ClippedAdam.
 * The same kernels and arithmetic as the host loop -- the same numbers.  step0 = the value step_dev holds when the call is made
 * (the host's mirror: used for the non-finite-loss latch).  grad_acc: DEVICE float[total], caller-owned (kept in the signature;
 * not written since the particles have buffers of their own).  At most 16 particles (VC_ERR_UNSUPPORTED beyond).  Cells sharded:
 * VC_ERR_STATE (the average has to cross the all-reduce before the optimiser; the host loop does that). */
int vc_svi_run_particles(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, int64_t step0, float* grad,
                         float* grad_acc, double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr,
                         double lrd, double beta1, double beta2, double adam_eps, double clip_norm, int num_particles,
                         int64_t n_steps, void* hip_stream);

/* --- cells sharded over ranks: the fused step cut at its one exchange ------------------------------
 * The per-step sequence of a rank is  K_main -> phase A -> [sum of the exchange buffer over all ranks] -> phase B  (three
 * launches + the exchange; the single-rank step of vc_svi_run_fused is the same code with nothing to sum).  It replaces, for
 * `svi.step` under a process group (velocity_inference_model.py:118-121 has no multi-GPU form: SURVEY.md F1), the
 * five-kernel sequence vc_elbo_grad + all-reduce + vc_clipped_adam.
 *
 * Exchange buffer `xbuf`: DEVICE float[vc_exchange_size], caller-owned like params / grad, zero-initialised once.
 *   [0, header + n_global)                         gradient partials of the replicated parameters, offsets of `grad`
 *   [pw_off, pw_off + pw_cap * Nx * Nhw)           per-cell-block partials of d loglik / d nu_omega (rows beyond a rank's
 *                                                  own cell blocks are zero), pw_cap = ceil(ceil(Nc_global / world) / 256)
 *   [loss_off, loss_off + 4 * (1 + ceil(Ng/64)))   the rank's loss terms, each double as four floats on fixed grids (their
 *                                                  float32 sum over <= 16 ranks is exact)
 * Every element is additive over ranks; replicated prior / entropy terms are contributed by rank 0 only.  After the sum
 * every rank holds the complete gradient and applies the identical update: parameters stay replicated bit for bit.
 *
 * phase = VC_PHASE_A : K_main + phase A of ONE step (n_steps must be 1); the caller then sums xbuf over the ranks
 *                      (e.g. torch.distributed.all_reduce -- any backend) on the same stream and calls
 * phase = VC_PHASE_B : phase B of that step (optimiser on the summed gradient, next guide sample, loss of the step into
 *                      loss_dev[t % loss_slots], step_dev = t + 1 was already set by phase A's K_main).
 * phase = VC_PHASE_AB: n_steps whole steps enqueued from this one call, the exchange made by the engine's own communicator
 *                      (vc_comm_init_rccl: ncclAllReduce on hip_stream between the two phases; world_size 1 needs none).
 * prime: as vc_svi_run_fused (sampling is rank-local: no exchange).  Same arguments otherwise. */
#define VC_PHASE_A 1
#define VC_PHASE_B 2
#define VC_PHASE_AB 3
int vc_exchange_size(const vc_engine* e, int64_t* n_floats);
int vc_svi_run_sharded(vc_engine* e, float* params, uint64_t seed, int64_t* step_dev, float* grad, float* xbuf,
                       double* loss_dev, int64_t loss_slots, float* exp_avg, float* exp_avg_sq, double lr, double lrd,
                       double beta1, double beta2, double adam_eps, double clip_norm, int prime, int phase,
                       int64_t n_steps, void* hip_stream);
/* The engine's own communicator for VC_PHASE_AB: RCCL, loaded at run time from `rccl_path` (the librccl.so the process
 * already uses, e.g. the one bundled with PyTorch -- the library itself does not link RCCL).  Rank 0 creates the 128-byte
 * unique id (vc_comm_rccl_unique_id), the host side broadcasts it (any transport), every rank of the engine's world_size
 * calls vc_comm_init_rccl with it (collective).  vc_destroy releases the communicator. */
int vc_comm_rccl_unique_id(const char* rccl_path, void* id_out_128_bytes);
int vc_comm_init_rccl(vc_engine* e, const char* rccl_path, const void* id_128_bytes);
/* Sum of a DEVICE float buffer over the ranks of that communicator, in place, on hip_stream (the collective of VC_PHASE_AB as a
 * call of its own: the host side uses it once to check the communicator against torch.distributed's sum of the same buffer). */
int vc_comm_allreduce(vc_engine* e, float* buf, int64_t n, void* hip_stream);

/* The one-shot exchange for VC_PHASE_AB over peer-mapped device memory (SURVEY.md section 5's latency-optimised variant;
 * velocycle_amd/csrc/vc_p2p_exchange.hip): every rank publishes its exchange buffer in a region of its own HBM and a
 * step-stamped flag in every peer's region, then reads all N buffers directly and adds them in fixed rank order (identical
 * bits on every rank, no float atomics).  vc_p2p_alloc creates this rank's region and returns its 64-byte hipIpcMemHandle_t;
 * the host side gathers the handles of all ranks in rank order (any transport) and hands the world_size x 64 bytes to
 * vc_p2p_connect (collective in effect: every rank must do it before the first step).  When connected, VC_PHASE_AB uses
 * this exchange instead of RCCL -- since round 6 without a launch of its own: phase B's blocks run the publish / wait protocol
 * and add the ranks' buffers in rank order where they read them (vc_tuning.p2p_separate = 1: the separate exchange kernel).  A peer that never publishes is detected by a bounded wait (vc_tuning.p2p_timeout_s, default 2 s):
 * vc_get_status then returns VC_ERR_STATE.  Opt-in: it has run across processes on one device only. */
int vc_p2p_alloc(vc_engine* e, void* ipc_handle_out_64_bytes);
int vc_p2p_connect(vc_engine* e, const void* all_handles_world_x_64_bytes);

/* One draw of the guide pushed through the deterministic part of the model (what
 * `Predictive(model, guide=guide, num_samples=1)` evaluates for the latent and deterministic sites;
 * velocity_inference_model.py:279-291, phase_inference_model.py:274-302): runs the sampling kernel only, no
 * likelihood.  The site values are then available through vc_read_site. */
int vc_sample_guide(vc_engine* e, const float* params, const float* eps, uint64_t seed, int64_t step,
                    void* hip_stream);

/* n_draws guide draws pushed through the deterministic part of the model, batched on the device: what
 * `Predictive(model, guide=guide, num_samples=n_draws)` evaluates for the latent and deterministic sites
 * (velocity_inference_model.py:279-291 / posterior_sampling :189-262, phase_inference_model.py:274-302).  Draw i
 * uses the Philox stream (seed, step0 + i), i.e. it equals vc_sample_guide(e, params, NULL, seed, step0 + i).
 * sites[k] is a VC_SITE_* id or VC_DET_PHI / VC_DET_OMEGA; out_dev[k] is DEVICE memory, float
 * [n_draws][length of that site], filled draw-major.  Asynchronous on hip_stream: nothing is copied to the host. */
int vc_sample_posterior(vc_engine* e, const float* params, uint64_t seed, int64_t step0, int64_t n_draws,
                        int n_sites, const int* sites, float* const* out_dev, void* hip_stream);

/* The dense E[log S] / E[log U] summaries that posterior_sampling adds to the posterior
 * (velocity_inference_model.py:236-258, phase_inference_model.py:248-262), evaluated on the device:
 *   out_S [g][c] = nu[g,:] . zeta(phi_c) + sum_b Db[b,c] dnu[b,g] + count_factor[c]
 *   out_S2[g][c] = the same with the constant count factor cf_avg
 *   out_U [g][c] = -logbeta[g] + log(relu(nu[g,:] . zeta'(phi_c) * omega[c] + gamma[g]) + 1e-5) + out_S[g][c]
 *   out_U2       = the same on out_S2
 * Every pointer is DEVICE memory: nu float[Ng][Nh], dnu float[Nb][Ng] (NULL without batches), phi / omega float[Nc_local],
 * logbeta / gamma float[Ng]; outputs float[Ng][Nc_local] row-major.  omega, logbeta, gamma, out_U, out_U2 are NULL for the
 * phase model.  count_factor and Db are the ones handed to vc_set_cell_data.  Asynchronous on hip_stream. */
int vc_expected_logs(vc_engine* e, const float* nu, const float* dnu, const float* phi, const float* omega,
                     const float* logbeta, const float* gamma, float cf_avg, float* out_S, float* out_S2,
                     float* out_U, float* out_U2, void* hip_stream);

/* introspection ----------------------------------------------------------------------------- */
/* Copies the value a site took in the last vc_elbo_grad to host memory (synchronises the stream). */
int vc_read_site(vc_engine* e, int site, float* host_out, int64_t n, void* hip_stream);
int vc_get_stats(const vc_engine* e, vc_stats* out);
/* The per-gene count histograms vc_finalize built (CSR over [S genes..., U genes...]: ptr int32[2*Ng + 1], distinct
 * non-zero count values and their multiplicities) -- the sufficient statistic of the negative binomial's lgamma /
 * digamma terms.  Pass NULL arrays to query *n_entries first.  Lets tests hold the device-built histograms against the
 * host pass (vc_tuning.host_hist). */
int vc_get_histogram(const vc_engine* e, int64_t* n_entries, int32_t* ptr_out, float* val_out, float* cnt_out);
/* Failure detection (the reference's counterpart: pyro.util.warn_if_nan(loss, "loss") inside SVI.step, call sites
 * phase_inference_model.py:169 / velocity_inference_model.py:120).  The last kernel of every step checks this rank's
 * loss on the device and latches the first step whose loss was NaN / Inf, so the check costs no host round trip per
 * step.  Synchronises `hip_stream`, returns VC_OK, or VC_ERR_NONFINITE with *first_bad_step (may be NULL) = the
 * 0-based index of the first such step and *n_bad (may be NULL) = how many steps were affected since vc_finalize /
 * the last vc_clear_status. */
int vc_get_status(vc_engine* e, int64_t* first_bad_step, int64_t* n_bad, void* hip_stream);
int vc_clear_status(vc_engine* e, void* hip_stream);
/* Kernel timing for bench.py's roofline: while enabled, every vc_elbo_grad brackets the likelihood
 * kernel with a pair of hipEvents recorded on the launch stream (not capturable into a hipGraph).
 * vc_get_timing synchronises the pending events and returns the accumulated duration (ms) and the
 * number of launches since timing was enabled. */
/* Shader clock the device is running at right now, measured on the device: one wave spins for `window_us` of the
 * constant 100 MHz wall clock (s_memrealtime) and counts shader-clock ticks (s_memtime).  Lets bench.py tell a
 * clock that has not ramped from a slower kernel.  Synchronises `hip_stream`. */
int vc_device_clock_mhz(double window_us, double* mhz_out, void* hip_stream);
int vc_set_timing(vc_engine* e, int enable);
int vc_get_timing(vc_engine* e, double* main_ms_total, int64_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* VELOCYCLE_HIP_H */
