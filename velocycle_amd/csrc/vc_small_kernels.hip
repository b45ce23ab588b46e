// O(Ng + Nc) kernels around the likelihood kernel:
//   vc_pack_counts   one-time re-layout of a strided (gene, cell) matrix into [gene block][cell][256]
//   vc_pre           guide sampling (reparameterisation), site values, prior / guide log-probs,
//                    gene table + cell table for K_main
//   vc_hist          negative-binomial lgamma / digamma terms from per-gene count histograms (fp64)
//   vc_post_gene     second-stage reduction of gene-level partials + chain rule to parameter gradients
//   vc_post_cell     second-stage reduction of cell-level partials, atan2 Jacobian, omega partials
//   vc_fin           loss assembly in fp64, angular-speed gradients, loss header
// Reference semantics restated: velocity_inference_guide.py:9-141, phase_inference_guide.py:10-56,
// priors of velocity_inference_model.py:322-353,383 / phase_inference_model.py:360-366,392.
// No compiler-chosen fused multiply-adds in this translation unit: the same source statement has to give the same bits in every
// kernel it is inlined into (K_post / K_fin, K_tail / K_omega, the sharded phases, the merged tail launch) -- hipcc's contraction
// of a * b + c depends on the surroundings of the statement.  Where a fused operation is wanted it is written as fmaf().
#pragma clang fp contract(off)
#include "vc_common.h"
#include "vc_tail_spec.h"
#include "vc_host_logic.h"     // VC_HIST_CAP


// ---------------------------------------------------------------------------------------------
// clock probe: shader-clock ticks (s_memtime) elapsed while the constant 100 MHz wall clock (s_memrealtime) advances
// by `wall_ticks`
__global__ void vc_clock_probe_kernel(unsigned long long wall_ticks, unsigned long long* out2) {
  const unsigned long long w0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
  unsigned long long w = w0;
  while (w - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(8); w = __builtin_amdgcn_s_memrealtime(); }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out2[0] = c1 - c0; out2[1] = w - w0; }
}

void vc_launch_clock_probe(unsigned long long wall_ticks, unsigned long long* out2, hipStream_t st) {
  hipLaunchKernelGGL(vc_clock_probe_kernel, dim3(1), dim3(64), 0, st, wall_ticks, out2);
}

// ---------------------------------------------------------------------------------------------
// Per-gene count histogram entry of one matrix element, built on the device while the matrix is re-laid-out (so that no
// dense matrix ever crosses back to the host): counts 1 .. VC_HIST_CAP-1 increment a dense per-gene bin (integer atomics:
// order-independent, deterministic); the rare other non-zero values (>= VC_HIST_CAP, non-integer) go to an overflow list
// that the host merges; negative / NaN / infinite values raise a flag (vc_finalize -> VC_ERR_ARG).
struct VcHistDev {
  unsigned* tab;            // [Ng][VC_HIST_CAP], nullptr: no histogram wanted (Lognormal noise)
  float* ovf_val;           // overflow entries
  int* ovf_gene;
  unsigned* ovf_n;          // entries appended (may exceed ovf_cap: then the host falls back to its own pass)
  unsigned ovf_cap;
  int* bad;                 // [0] invalid count value seen, [1] invalid CSR index seen, [2] a count that uint16 cannot hold
};

__device__ __forceinline__ void vc_hist_put(const VcHistDev& h, int g, float v) {
  if (!(v >= 0.f && v <= 3.0e38f)) { h.bad[0] = 1; return; }
  if (v == 0.f) return;
  if (v > 65535.f || (float)(int)v != v) h.bad[2] = 1;
  if (!h.tab) return;
  if (v < (float)VC_HIST_CAP && (float)(int)v == v) atomicAdd(&h.tab[(size_t)g * VC_HIST_CAP + (int)v], 1u);
  else {
    const unsigned i = atomicAdd(h.ovf_n, 1u);
    if (i < h.ovf_cap) { h.ovf_val[i] = v; h.ovf_gene[i] = g; }
  }
}

// ord (or nullptr): the cell stored at position c of the blocked layout (cells ordered by batch: vc_order_by_batch)
__global__ void vc_pack_counts_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                      long long gs, long long cs, int Ng, int Nc, int nGB, int gbw, int log1p_t,
                                      VcHistDev h, const int* __restrict__ ord) {
  const long long total = (long long)nGB * Nc * gbw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int gl = (int)(i % gbw);
    const long long t = i / gbw;
    const int c = (int)(t % Nc);
    const int gb = (int)(t / Nc);
    const int g = gb * gbw + gl;
    float v = 0.f;
    if (g < Ng) {
      v = src[(long long)g * gs + (long long)(ord ? ord[c] : c) * cs];
      vc_hist_put(h, g, v);
      if (log1p_t) v = (float)log((double)v + 1.0 + 1e-16);   // preprocessing.py:154 / :267
    }
    dst[i] = v;
  }
}

// float32 blocked counts -> uint16 (called only when every count is an integer <= 65535: exact)
__global__ void vc_counts_to_u16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const float4 v = *reinterpret_cast<const float4*>(src + i);
    ushort4 o;
    o.x = (unsigned short)v.x; o.y = (unsigned short)v.y; o.z = (unsigned short)v.z; o.w = (unsigned short)v.w;
    *reinterpret_cast<ushort4*>(dst + i) = o;
  }
}
void vc_launch_counts_to_u16(const float* src, unsigned short* dst, long long n, hipStream_t st) {
  hipLaunchKernelGGL(vc_counts_to_u16_kernel, dim3(4096), dim3(256), 0, st, src, dst, n);      // n is a multiple of 256
}

// CSR (cells x genes, canonical: no duplicate entries) -> the blocked layout, one wave per cell; dst is pre-zeroed.
// Reference counterpart: the `.A` / `.toarray()` densification of preprocessing.py:141-147, 243-252.
// pos (or nullptr): the position of cell c in the blocked layout
__global__ __launch_bounds__(256) void vc_scatter_csr_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                             const float* __restrict__ data, float* __restrict__ dst, int Ng,
                                                             int Nc, int gbw, int log1p_t, VcHistDev h, const int* __restrict__ pos) {
  const int lane = threadIdx.x & 63;
  for (long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); c < Nc; c += (long long)gridDim.x * 4) {
    const long long beg = indptr[c], end = indptr[c + 1];
    for (long long k = beg + lane; k < end; k += 64) {
      const int g = indices[k];
      if (g < 0 || g >= Ng) { h.bad[1] = 1; continue; }
      float v = data[k];
      vc_hist_put(h, g, v);
      if (log1p_t) v = (float)log((double)v + 1.0 + 1e-16);
      dst[((size_t)(g / gbw) * Nc + (pos ? pos[c] : c)) * gbw + (g % gbw)] = v;
    }
  }
}

static VcHistDev vc_hist_dev(unsigned* tab, float* ovf_val, int* ovf_gene, unsigned* ovf_n, unsigned ovf_cap, int* bad) {
  VcHistDev h;
  h.tab = tab; h.ovf_val = ovf_val; h.ovf_gene = ovf_gene; h.ovf_n = ovf_n; h.ovf_cap = ovf_cap; h.bad = bad;
  return h;
}

void vc_launch_pack_counts(const float* src, float* dst, long long gene_stride, long long cell_stride,
                           int Ng, int Nc, int nGB, int gbw, int log1p_transform, unsigned* tab, float* ovf_val,
                           int* ovf_gene, unsigned* ovf_n, unsigned ovf_cap, int* bad, const int* ord, hipStream_t st) {
  hipLaunchKernelGGL(vc_pack_counts_kernel, dim3(2048), dim3(256), 0, st, src, dst, gene_stride,
                     cell_stride, Ng, Nc, nGB, gbw, log1p_transform, vc_hist_dev(tab, ovf_val, ovf_gene, ovf_n, ovf_cap, bad), ord);
}

void vc_launch_scatter_csr(const long long* indptr, const int* indices, const float* data, float* dst, int Ng, int Nc,
                           int gbw, int log1p_transform, unsigned* tab, float* ovf_val, int* ovf_gene, unsigned* ovf_n,
                           unsigned ovf_cap, int* bad, const int* pos, hipStream_t st) {
  int nb = (Nc + 3) / 4;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(vc_scatter_csr_kernel, dim3(nb), dim3(256), 0, st, indptr, indices, data, dst, Ng, Nc, gbw,
                     log1p_transform, vc_hist_dev(tab, ovf_val, ovf_gene, ovf_n, ovf_cap, bad), pos);
}

// ---------------------------------------------------------------------------------------------
// K_hist (NB): one wave per task (lists of distinct values) or one block per gene block (dense tables), fp64 -- also runs as
// extra blocks of K_pre
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vc_hist_kernel(const VcDims d, const VcBufs b,
                                                      const float* __restrict__ P, int cond_only) {
  if (d.hist_dense) {
    vc_hist_dense_quarter(d, b, blockIdx.x >> 2, blockIdx.x & 3, P, cond_only, 0, vc_hist_lds());
    return;
  }
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (task < b.n_tasks) vc_hist_wave(d, b, P, cond_only, task, threadIdx.x & 63);
}

void vc_launch_hist(const VcDims& d, const VcBufs& b, const float* params, int cond_only, hipStream_t st) {
  hipLaunchKernelGGL(vc_hist_kernel, dim3(vc_hist_blocks_pre(d, b)), dim3(256), vc_hist_dyn_lds(d, 1, 256), st, d, b, params, cond_only);
}

// ---------------------------------------------------------------------------------------------
// K_pre
// ---------------------------------------------------------------------------------------------
// MULTI (vc_svi_run_particles): ONE launch draws the samples of all K particles of a step -- blockIdx.y = particle, whose
// workspaces are entry blockIdx.y of `bs` (device array; the parameters are the same for all of them)
template <bool MULTI, int SPEC = 0>
__global__ __launch_bounds__(256) void vc_pre_kernel(const VcDims d, const VcBufs b0, const VcBufs* __restrict__ bs,
                                                     const float* __restrict__ P,
                                                     const float* __restrict__ eps_in, uint64_t seed,
                                                     long long step_host,
                                                     const long long* __restrict__ step_dev,
                                                     int cond_only, int particles, int particle_in) {
  vc_spec_assume<SPEC>(d);      // (vc_tail_spec.h: the K-particle step of a configuration with a compiled signature)
  if (SPEC > 0) { __builtin_assume(cond_only == 0); __builtin_assume(eps_in == nullptr); }
  __shared__ double sm_red[16];
  __shared__ float s_nuw[VC_MAX_NW];
  VcBufs bm;
  if (MULTI) bm = bs[blockIdx.y];
  const VcBufs& b = MULTI ? bm : b0;
  const int particle = MULTI ? (int)blockIdx.y : particle_in;
  VC_KSTAMP(0, 0);
  // Philox stream of this draw: (seed, step); with K particles per step the k-th draw of step t is stream t K + k
  const long long step = (step_dev ? *step_dev : step_host) * particles + particle;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  const bool nb = d.noise == VC_NOISE_NB;
  double loss = 0.0;

  if ((int)blockIdx.x >= d.nb_pre_gene + d.nb_pre_cell) {
    // ------------------------------- histogram part (NB) -------------------------------------------
    const int hb = blockIdx.x - d.nb_pre_gene - d.nb_pre_cell;
    if (d.hist_dense) {          // four quarter blocks per gene block, the dense tail-count tables
      vc_hist_dense_quarter(d, b, hb >> 2, hb & 3, P, cond_only, 0, vc_hist_lds());
      return;
    }
    const int task = hb * 4 + (threadIdx.x >> 6);          // one wave per task of <= 64 distinct values
    if (task < b.n_tasks) vc_hist_wave(d, b, P, cond_only, task, threadIdx.x & 63);
    return;
  }
  if ((int)blockIdx.x < d.nb_pre_gene) {
    // ------------------------------- gene part ------------------------------------------------
    // 64 genes per block (lane = gene); the sites of a gene are independent ROLES spread over the 4 waves, so that the
    // serial path of a thread is one or two counter-RNG draws instead of the Nh + 2 of a whole gene:
    //   role 0: log gamma / log beta (+ rho); 1 .. Nh: nu[h]; Nh + 1 .. Nh + Nb: delta nu; last: shape_inv.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + lane;
    // LRMN: eps_W is the same for every gene: lane k of wave 0 draws eps_W[k] once, with all lanes active (the
    // padded lanes of the last block included), and the role reads it with v_readlane
    float ew_mine = 0.f;
    if (wv == 0 && lrmn && !cond_only && lane < d.R)
      ew_mine = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_LRMN_W] + lane, d.eoff[VC_E_LRMN_W] + lane);
    // broadcast to wave-uniform values NOW, while every lane is active: after the divergence below the registers of the
    // lanes that take the other path are free for the compiler to reuse
    float ew_all[VC_MAX_RANK];
#pragma unroll
    for (int k = 0; k < VC_MAX_RANK; ++k)
      ew_all[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ew_mine), k));
    if (g < d.Ng_pad) {
      float* GT = b.GT + g;
      const size_t NP = d.Ng_pad;
      if (g >= d.Ng) {   // padded gene: nu~ = 0 (never reaches a per-cell sum), loss masked in K_main
        if (wv == 0) {
          for (int k = 0; k < d.K; ++k) GT[k * NP] = 0.f;
          GT[d.K * NP] = 0.f; GT[(d.K + 1) * NP] = 1.f; GT[(d.K + 2) * NP] = 1.f;
        }
      } else {
        float logp = 0.f, logq = 0.f;
        const int nroles = d.Nh + d.Nb + 2;
        // role 0 (two or more RNG draws) has wave 0 to itself; roles 1..3 go to waves 1..3, the rest round-robin over them
        for (int role = 0; role < nroles; ++role) {
          if ((role < 4 ? role : 1 + (role - 4) % 3) != wv) continue;
          if (role >= 1 && role <= d.Nh) {
            // ---- nu[h] ----
            const int h = role - 1;
            const long long j = (long long)g * d.Nh + h;
            float x;
            if (cond_only) {
              x = CND(VC_SITE_NU) ? b.cnd[VC_SITE_NU][j] : 0.f;
            } else {
              const float e = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_NU] + j, d.eoff[VC_E_NU] + j);
              const float u = P[d.poff[VC_P_NU_USCALES] + j];
              const float xg = P[d.poff[VC_P_NU_LOCS] + j] + expf(u) * e;
              if (CND(VC_SITE_NU)) x = b.cnd[VC_SITE_NU][j];
              else { x = xg; logq += -0.5f * e * e - u - 0.5f * VC_LOG_2PI; }
              logp += vc_normal_lp(x, b.mu_nu[j], b.sd_nu[j]);
              b.lat[VC_SITE_NU][j] = x;
            }
            GT[h * NP] = x;
          } else if (role > d.Nh && role <= d.Nh + d.Nb) {
            // ---- delta nu (Delta guide) ----
            const int q = role - d.Nh - 1;
            const long long j = (long long)q * d.Ng + g;
            float x = 0.f;
            if (d.with_dnu) {
              if (cond_only) x = CND(VC_SITE_DNU) ? b.cnd[VC_SITE_DNU][j] : 0.f;
              else {
                x = CND(VC_SITE_DNU) ? b.cnd[VC_SITE_DNU][j] : P[d.poff[VC_P_DNU_LOCS] + j];
                logp += vc_normal_lp(x, 0.f, vel ? 0.01f : b.sd_dnu[j]);
                b.lat[VC_SITE_DNU][j] = x;
              }
              GT[(d.Nh + q) * NP] = x;
            }
          } else if (role == nroles - 1) {
            // ---- shape_inv (Delta guide, positive) ----
            float si = 1.f;
            if (nb) {
              if (cond_only) si = CND(VC_SITE_SHAPE_INV) ? b.cnd[VC_SITE_SHAPE_INV][g] : 1.f;
              else {
                si = CND(VC_SITE_SHAPE_INV) ? b.cnd[VC_SITE_SHAPE_INV][g] : expf(P[d.poff[VC_P_SHAPE_INV_ULOCS] + g]);
                logp += d.gamma_alpha * logf(d.gamma_beta) + (d.gamma_alpha - 1.f) * logf(si) -
                        d.gamma_beta * si - d.lgamma_alpha;
                b.lat[VC_SITE_SHAPE_INV][g] = si;
              }
            }
            GT[(d.K + 2) * NP] = 1.0f / si;
          } else {
            // ---- role 0: log gamma, log beta ----
            float lg = 0.f, lbv = 0.f;
            if (vel && !cond_only) {
              float lg_guide, lb_guide;
              if (!lrmn) {
                const float eg = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_LOGGAMMA] + g, d.eoff[VC_E_LOGGAMMA] + g);
                const float eb = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_LOGBETA] + g, d.eoff[VC_E_LOGBETA] + g);
                const float ug = P[d.poff[VC_P_LOGGAMMA_USCALES] + g], ub = P[d.poff[VC_P_LOGBETA_USCALES] + g];
                lg_guide = P[d.poff[VC_P_LOGGAMMA_LOCS] + g] + expf(ug) * eg;
                lb_guide = P[d.poff[VC_P_LOGBETA_LOCS] + g] + expf(ub) * eb;
                if (!CND(VC_SITE_LOGGAMMA)) logq += -0.5f * eg * eg - ug - 0.5f * VC_LOG_2PI;
                if (!CND(VC_SITE_LOGBETA)) logq += -0.5f * eb * eb - ub - 0.5f * VC_LOG_2PI;
              } else {
                // LowRankMultivariateNormal.rsample: X = loc + W eps_W + sqrt(cov_diag) eps_D
                // columns 0..3 and 4..7 are summed separately and then added: the association the fused step uses
                // (vc_fused_kernels.hip splits the cov_factor row over two roles), so both paths draw bit-identical samples
                float delta = 0.f, w2 = 0.f, delta_hi = 0.f, w2_hi = 0.f;
#pragma unroll
                for (int k = 0; k < VC_MAX_RANK; ++k)
                  if (k < d.R) {
                    const float w = expf(P[d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)g * d.R + k]);
                    if (k < 4) { delta += w * ew_all[k]; w2 += w * w; }
                    else { delta_hi += w * ew_all[k]; w2_hi += w * w; }
                  }
                if (d.R > 4) { delta += delta_hi; w2 += w2_hi; }
                const float dg = expf(P[d.poff[VC_P_LRMN_UCOV_DIAG] + g]);
                const float ed = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_LRMN_D] + g, d.eoff[VC_E_LRMN_D] + g);
                delta += sqrtf(dg) * ed;
                const float sgam = sqrtf(w2 + dg);
                lg_guide = P[d.poff[VC_P_LRMN_LOC] + g] + delta;
                const float rho_real_g = P[d.poff[VC_P_RHO_REAL_LOC] + g];
                const float rho = sigmoidf_(rho_real_g / d.rho_scale) * 1.998f - 0.999f;
                const float ub = P[d.poff[VC_P_LOGBETA_USCALES] + g];
                const float sb = expf(ub);
                const float eb = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_LOGBETA] + g, d.eoff[VC_E_LOGBETA] + g);
                const float tt = sb * sqrtf(1.f - rho * rho);
                lb_guide = P[d.poff[VC_P_LOGBETA_LOCS] + g] + rho * sb * delta / sgam + tt * eb;
                if (!CND(VC_SITE_LOGBETA)) logq += -0.5f * eb * eb - logf(tt) - 0.5f * VC_LOG_2PI;
                b.lat_delta[g] = delta;
                b.lat_sgam[g] = sgam;
                const float rho_val = CND(VC_SITE_RHO_REAL) ? b.cnd[VC_SITE_RHO_REAL][g] : rho_real_g;
                logp += vc_normal_lp(rho_val, d.rho_mean, d.rho_std);
                b.lat[VC_SITE_RHO_REAL][g] = rho_val;
              }
              lg = CND(VC_SITE_LOGGAMMA) ? b.cnd[VC_SITE_LOGGAMMA][g] : lg_guide;
              lbv = CND(VC_SITE_LOGBETA) ? b.cnd[VC_SITE_LOGBETA][g] : lb_guide;
              logp += vc_normal_lp(lg, b.mu_g[g], b.sd_g[g]) + vc_normal_lp(lbv, b.mu_b[g], b.sd_b[g]);
              b.lat[VC_SITE_LOGGAMMA][g] = lg;
              b.lat[VC_SITE_LOGBETA][g] = lbv;
            }
            GT[d.K * NP] = lbv;
            GT[(d.K + 1) * NP] = expf(lg);
          }
        }
        loss = -(double)d.root_w * ((double)logp - (double)logq);
      }
    }
  } else {
    // ------------------------------- cell part ------------------------------------------------
    const int bc = blockIdx.x - d.nb_pre_gene;
    if (vel && (int)threadIdx.x < d.NW) {
      const int j = threadIdx.x;
      float val = 0.f, lq = 0.f;
      if (!cond_only) {
        if (!lrmn) {
          const float e = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_NUOMEGA] + j, d.eoff[VC_E_NUOMEGA] + j);
          const float u = P[d.poff[VC_P_NUOMEGA_USCALES] + j];
          val = P[d.poff[VC_P_NUOMEGA_LOCS] + j] + expf(u) * e;
          lq = -0.5f * e * e - u - 0.5f * VC_LOG_2PI;
        } else {
          const long long i = (long long)d.Ng + j;
          float delta = 0.f;
          for (int k = 0; k < d.R; ++k) {
            const float ew = eps_in ? eps_in[d.eoff[VC_E_LRMN_W] + k]
                                    : vc_philox_normal(seed, step, d.eoff[VC_E_LRMN_W] + k);
            delta += expf(P[d.poff[VC_P_LRMN_UCOV_FACTOR] + i * d.R + k]) * ew;
          }
          const float ed = vc_eps(eps_in, b.eps_used, seed, step, d.eoff[VC_E_LRMN_D] + i, d.eoff[VC_E_LRMN_D] + i);
          delta += sqrtf(expf(P[d.poff[VC_P_LRMN_UCOV_DIAG] + i])) * ed;
          val = P[d.poff[VC_P_LRMN_LOC] + i] + delta;
          if (bc == 0) b.lat_delta[i] = delta;
        }
        const float x = CND(VC_SITE_NUOMEGA) ? b.cnd[VC_SITE_NUOMEGA][j] : val;
        if (bc == 0) {
          b.lat[VC_SITE_NUOMEGA][j] = x;
          const float lp = vc_normal_lp(x, b.mu_w[j], b.sd_w[j]);
          loss = -(double)d.root_w * ((double)lp - ((CND(VC_SITE_NUOMEGA) || lrmn) ? 0.0 : (double)lq));
        }
        val = x;
      }
      s_nuw[j] = val;
    }
    __syncthreads();
    const int c = bc * 256 + threadIdx.x;
    if (c < d.Nc) {
      float x, y;
      if (cond_only) {
        x = CND(VC_SITE_PHIXY) ? b.cnd[VC_SITE_PHIXY][2 * c] : 1.f;
        y = CND(VC_SITE_PHIXY) ? b.cnd[VC_SITE_PHIXY][2 * c + 1] : 0.f;
      } else {
        const long long li = d.eoff[VC_E_PHIXY] + 2LL * c;
        const long long gi = d.eoff[VC_E_PHIXY] + 2LL * (d.cell_offset + c);
        const float ex = vc_eps(eps_in, b.eps_used, seed, step, li, gi);
        const float ey = vc_eps(eps_in, b.eps_used, seed, step, li + 1, gi + 1);
        const float px = b.pxy[2 * c], py = b.pxy[2 * c + 1];
        if (CND(VC_SITE_PHIXY)) {
          x = b.cnd[VC_SITE_PHIXY][2 * c]; y = b.cnd[VC_SITE_PHIXY][2 * c + 1];
          loss += 0.5 * ((double)(x - px) * (x - px) + (double)(y - py) * (y - py)) + (double)VC_LOG_2PI;
        } else {
          x = P[d.poff[VC_P_PHIXY_LOCS] + 2LL * c] + ex;
          y = P[d.poff[VC_P_PHIXY_LOCS] + 2LL * c + 1] + ey;
          // -(log p - log q): the -log(2 pi) of prior and guide cancel
          loss += 0.5 * ((double)(x - px) * (x - px) + (double)(y - py) * (y - py)) -
                  0.5 * ((double)ex * ex + (double)ey * ey);
        }
        b.lat[VC_SITE_PHIXY][2 * c] = x;
        b.lat[VC_SITE_PHIXY][2 * c + 1] = y;
      }
      const float phi = atan2f(y, x);                      // utils.py:505-506 (the deterministic site; the basis does not go through it)
      float s1, c1;
      vc_dir_sincos(x, y, &s1, &c1);
      float sk[VC_MAXH], ck[VC_MAXH];
      sk[0] = s1; ck[0] = c1;
      const int hm = d.H > d.Hw ? d.H : d.Hw;
      for (int k = 1; k < hm && k < VC_MAXH; ++k) {
        sk[k] = sk[k - 1] * c1 + ck[k - 1] * s1;
        ck[k] = ck[k - 1] * c1 - sk[k - 1] * s1;
      }
      float omega = 0.f, domega = 0.f;
      if (vel && !cond_only) {
        for (int xq = 0; xq < d.Nx; ++xq) {
          const float* nw = s_nuw + xq * d.Nhw;
          float o = nw[0], dd = 0.f;
          for (int k = 0; k < d.Hw; ++k) {
            o += nw[2 * k + 1] * sk[k] + nw[2 * k + 2] * ck[k];
            dd += (float)(k + 1) * (nw[2 * k + 1] * ck[k] - nw[2 * k + 2] * sk[k]);
          }
          const float dx = b.Dm[(size_t)xq * d.Nc + c];
          omega += dx * o;
          domega += dx * dd;
        }
      }
      // every value stored twice {x, x}: K_main fetches it as an SGPR pair = packed-math operand
      const int cp = vc_pos(b, c);                     // the record's place in the likelihood kernel's cell order
      float2* ct = reinterpret_cast<float2*>(b.CT + (size_t)cp * d.ctw);
      for (int k = 0; k < d.H; ++k) { ct[2 * k] = make_float2(sk[k], sk[k]); ct[2 * k + 1] = make_float2(ck[k], ck[k]); }
      for (int q = 0; q < d.nbk; ++q) {                // (none when the one-hot batch offsets are folded per workgroup)
        const float v = b.Dbm[(size_t)q * d.Nc + c];
        ct[2 * d.H + q] = make_float2(v, v);
      }
      const int nbk = d.nbk;
      vc_rec_put_omega(ct, d, omega, sk, ck);
      { const float cfs = b.cf[c] * vc_rec_cf_scale(d.noise); ct[2 * d.H + nbk + 1] = make_float2(cfs, cfs); }
      vc_put_w(d, b, c, cp, sk, ck);
      b.lat_phi[c] = phi;
      b.lat_omega[c] = omega;
      b.lat_domega[c] = domega;
    }
  }
  VC_KSTAMP(0, 2);
  const double tot = vc_block_sum_d(loss, sm_red);
  if (threadIdx.x == 0) b.LP[blockIdx.x] = cond_only ? 0.0 : tot;
  VC_KSTAMP(0, 3);
}

void vc_launch_pre(const VcDims& d, const VcBufs& b, const float* params, const float* eps,
                   uint64_t seed, long long step, const long long* step_dev, int cond_only, int with_hist,
                   hipStream_t st, int particles, int particle) {
  if (d.generic) { vc_launch_pre_generic(d, b, params, eps, seed, step, step_dev, cond_only, with_hist, st, particles, particle); return; }
  const int nb_hist = with_hist ? vc_hist_blocks_pre(d, b) : 0;
  hipLaunchKernelGGL(vc_pre_kernel<false>, dim3(d.nb_pre_gene + d.nb_pre_cell + nb_hist), dim3(256), vc_hist_dyn_lds(d, with_hist, 256), st, d, b,
                     (const VcBufs*)nullptr, params, eps, seed, step, step_dev, cond_only, particles, particle);
}
// the K particles of a step in one launch (fast kernel set only; b = the first particle's buffers: sizes and tables are common)
void vc_launch_pre_particles(const VcDims& d, const VcBufs& b, const VcBufs* bs_dev, const float* params, uint64_t seed,
                             const long long* step_dev, int with_hist, int K, hipStream_t st) {
  const int nb_hist = with_hist ? vc_hist_blocks_pre(d, b) : 0;
  const dim3 grid(d.nb_pre_gene + d.nb_pre_cell + nb_hist, K);
  if (vc_spec_launch<VC_SPECK_PARTICLES, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_pre_kernel<true, decltype(sp)::value>), grid, dim3(256), vc_hist_dyn_lds(d, with_hist, 256), st, d, b, bs_dev, params,
                           (const float*)nullptr, seed, 0LL, step_dev, 0, K, 0);
      }))
    return;
  hipLaunchKernelGGL(vc_pre_kernel<true>, grid, dim3(256), vc_hist_dyn_lds(d, with_hist, 256), st,
                     d, b, bs_dev, params, (const float*)nullptr, seed, 0LL, step_dev, 0, K, 0);
}

// K-particle step (vc_svi_run_particles): the average of the particles' gradients and losses, added up in particle order and
// multiplied by 1 / K (as the reference's Trace_ELBO averages loss and gradients over its particles:
// velocity_inference_model.py:79,111; x * (1 / K) is what a PyTorch division by a scalar computes), left in particle 0's buffer
// (header hi / lo = the averaged loss, also filed in the loss ring); advances the device step counter.
__global__ __launch_bounds__(256) void vc_particle_avg_kernel(const VcParticleGrads pg, long long n, double* __restrict__ loss_ring,
                                                             long long loss_slots, long long step, long long* __restrict__ step_dev) {
  const int K = pg.K;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double l = 0.0;
    for (int k = 0; k < K; ++k) l += (double)pg.g[k][0] + (double)pg.g[k][1];
    const double avg = l * (1.0 / (double)K);
    const float hi = (float)avg, lo = (float)(avg - (double)hi);
    pg.g[0][0] = hi; pg.g[0][1] = lo;
    if (loss_ring) loss_ring[loss_slots > 1 ? (step % loss_slots) : 0] = (double)hi + (double)lo;
    if (step_dev) step_dev[0] += 1;
  }
  const float inv = 1.0f / (float)K;
  for (long long i = 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float a = 0.f + pg.g[0][i];
    for (int k = 1; k < K; ++k) a += pg.g[k][i];
    pg.g[0][i] = a * inv;
  }
}
void vc_launch_particle_avg(const VcParticleGrads& pg, long long n, double* loss_ring, long long loss_slots, long long step,
                            long long* step_dev, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(vc_particle_avg_kernel, dim3((unsigned)nb), dim3(256), 0, st, pg, n, loss_ring, loss_slots, step, step_dev);
}

// ---------------------------------------------------------------------------------------------
// K_hist (NB): one wave per gene.  sum_k cnt_k (lgamma(r+k) - lgamma(r)) and its r-derivative in fp64.
// The lgamma terms of GammaPoisson.log_prob summed over cells depend on (r_g, k) only, so the per-gene
// histogram of the counts is a sufficient statistic -- no lgamma/digamma in the (gene, cell) loop.
// Depends on the parameters only, so it runs on a side stream concurrently with K_pre / K_main.
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// K_post_gene: 1024 threads = 16 waves; lanes = 64 genes; wave w sums chunks w, w+16, ...
// ---------------------------------------------------------------------------------------------
#define VC_PG_WAVES 16
#define VC_MAXQ (2 * VC_MAXH + 1 + VC_MAXNB + 3)
// MQ = compiled bound of the number of partial-sum rows (nq): 2 (U-only kernel), 6 (S+U kernel with H = 1, no batches;
// phase up to K = 5) or VC_MAXQ; the reduction loop keeps 4 x MQ loads in flight, so the small variants leave registers
// for the role inputs that are fetched ahead of it.

template <int MQ>
__device__ __forceinline__ void vc_post_gene_block(const VcDims& d, const VcBufs& b,
                                                   const float* __restrict__ P, float* __restrict__ G,
                                                   int gblock) {
  __shared__ float sm[VC_PG_WAVES][MQ][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = gblock * 64 + lane;
  const VcChunkWalk wk = vc_chunk_walk(d, b, g, wave);      // (one-hot batches: the waves walk their batch's chunks -- vc_common.h; a scalar
                                                            // load the reduction's addresses depend on: requested first)
  // Chain rule to the parameter gradients: the work of one gene is split into independent ROLES and every role runs
  // on its own wave (lane = gene), so that the serial path of the block is the longest role instead of their sum:
  // role h < Nh: nu[h]; Nh + q: dnu[q]; 12: shape_inv (+ r-only loss terms); 13: log gamma / log beta (mean-field) or
  // the LRMN core; 14: LRMN cov_factor row.
  // Everything a role reads besides the reduced sums (site values, priors, eps, parameters, histogram sums) does not
  // depend on K_main's partials: it is fetched HERE, ahead of the reduction loop, whose latency then covers these
  // (otherwise dependent) round trips.
  const int role = wave;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  const bool nb = d.noise == VC_NOISE_NB;
  const int K = d.Kq, KT = d.K, Nh = d.Nh;      // K: coefficient rows of K_main's partials; KT: rows of the gene table in front of log beta
  const float rw = d.root_w;
  // one-hot batches (d.onehot): the delta-nu gradients come from vc_dnu_range_sum, by the waves of roles Nh .. 11 in a loop
  // over their batches (any number of batches); the dense form keeps one role per batch
  const bool dnu_dense = d.with_dnu && !d.onehot;
  constexpr int NIN = 11 + VC_MAX_RANK;   // role 13 uses in[0..13], role 14 in[0..10] + the R cov_factor entries
  float in[NIN];
  double HLg = 0.0, HDg = 0.0;            // histogram task sums of this gene, fixed order
#pragma unroll
  for (int i = 0; i < NIN; ++i) in[i] = 0.f;
  if (g < d.Ng) {
    if (role < Nh) {
      if (!CND(VC_SITE_NU)) {
        const long long j = (long long)g * Nh + role;
        in[0] = b.lat[VC_SITE_NU][j]; in[1] = b.sd_nu[j]; in[2] = b.mu_nu[j];
        in[3] = b.eps_used[d.eoff[VC_E_NU] + j]; in[4] = P[d.poff[VC_P_NU_USCALES] + j];
      }
    } else if (role < Nh + d.Nb && dnu_dense) {
      if (!CND(VC_SITE_DNU)) {
        const long long j = (long long)(role - Nh) * d.Ng + g;
        in[0] = b.lat[VC_SITE_DNU][j]; in[1] = vel ? 0.01f : b.sd_dnu[j];
      }
    } else if (role == 12 && nb) {
      in[0] = b.GT[(size_t)(KT + 2) * d.Ng_pad + g];
      if (!CND(VC_SITE_SHAPE_INV)) in[1] = b.lat[VC_SITE_SHAPE_INV][g];
      for (int t = b.h_tptr[g]; t < b.h_tptr[g + 1]; ++t) { HLg += b.HL[t]; HDg += b.HD[t]; }
    } else if ((role == 13 || role == 14) && vel) {
      in[0] = b.GT[(size_t)(KT + 1) * d.Ng_pad + g];
      if (!CND(VC_SITE_LOGGAMMA)) { in[1] = b.lat[VC_SITE_LOGGAMMA][g]; in[2] = b.sd_g[g]; in[3] = b.mu_g[g]; }
      if (!CND(VC_SITE_LOGBETA)) { in[4] = b.lat[VC_SITE_LOGBETA][g]; in[5] = b.sd_b[g]; in[6] = b.mu_b[g]; }
      if (!lrmn) {
        if (role == 13) {
          in[7] = b.eps_used[d.eoff[VC_E_LOGGAMMA] + g]; in[8] = b.eps_used[d.eoff[VC_E_LOGBETA] + g];
          in[9] = P[d.poff[VC_P_LOGGAMMA_USCALES] + g]; in[10] = P[d.poff[VC_P_LOGBETA_USCALES] + g];
        }
      } else {
        in[7] = b.lat_delta[g]; in[8] = b.lat_sgam[g];
        in[9] = P[d.poff[VC_P_LOGBETA_USCALES] + g]; in[10] = P[d.poff[VC_P_RHO_REAL_LOC] + g];
        if (role == 13) {
          in[11] = b.eps_used[d.eoff[VC_E_LOGBETA] + g]; in[12] = P[d.poff[VC_P_LRMN_UCOV_DIAG] + g];
          in[13] = b.eps_used[d.eoff[VC_E_LRMN_D] + g];
        } else {
#pragma unroll
          for (int k = 0; k < VC_MAX_RANK; ++k)
            if (k < d.R) in[11 + k] = P[d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)g * d.R + k];
        }
      }
    }
  }

  // U chunk groups x MQ rows of loads are in flight per wave before any is consumed; U is chosen so that the usual
  // one-round grid of K_main (<= 16 U chunks) is fetched in a single round trip
  constexpr int U = MQ <= 2 ? 16 : (MQ <= 6 ? 8 : 4);
  float acc[MQ];
#pragma unroll
  for (int q = 0; q < MQ; ++q) acc[q] = 0.f;
  for (int ch0 = wk.first; ch0 < wk.end; ch0 += U * wk.stride) {
    float v[U][MQ];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ch = ch0 + u * wk.stride;
      const float* go = b.GO + ((size_t)(ch < wk.end ? ch : ch0) * d.nq) * d.Ng_pad + g;
#pragma unroll
      for (int q = 0; q < MQ; ++q) v[u][q] = (q < d.nq) ? go[(size_t)q * d.Ng_pad] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ch0 + u * wk.stride < wk.end) {
#pragma unroll
        for (int q = 0; q < MQ; ++q) acc[q] += v[u][q];
      }
  }
#pragma unroll
  for (int q = 0; q < MQ; ++q) sm[wave][q][lane] = acc[q];
  VC_KSTAMP(1, 1);
  __syncthreads();
  double loss = 0.0;
  auto T = [&](int q) {                 // reduced partial sum of output row q for this gene (fixed order)
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < VC_PG_WAVES; ++w) t += sm[w][q][lane];
    return t;
  };
  if (g < d.Ng) {
    if (role < Nh) {
      // ---- nu[h] ----
      const int h = role;
      const long long j = (long long)g * Nh + h;
      float gl = 0.f, gu = 0.f;
      if (!CND(VC_SITE_NU)) {
        const float x = in[0], sd = in[1];
        const float gx = T(h) - rw * (x - in[2]) / (sd * sd);
        gl = -gx;
        gu = -gx * expf(in[4]) * in[3] - rw;
      }
      G[d.poff[VC_P_NU_LOCS] + j] = gl;
      G[d.poff[VC_P_NU_USCALES] + j] = gu;
    } else if (role < Nh + d.Nb && dnu_dense) {
      // ---- delta nu[q] ----
      const int q = role - Nh;
      const long long j = (long long)q * d.Ng + g;
      float gl = 0.f;
      if (!CND(VC_SITE_DNU)) {
        const float x = in[0], sd = in[1];
        gl = -(T(Nh + q) - rw * x / (sd * sd));
      }
      G[d.poff[VC_P_DNU_LOCS] + j] = gl;
    } else if (role >= Nh && role < 12 && d.onehot) {
      // ---- delta nu[q], one-hot batches: q = role - Nh, + (12 - Nh), ... ; the likelihood part is the constant harmonic's
      // partial row summed over the batch's workgroups
      for (int q = role - Nh; q < d.Nb; q += 12 - Nh) {
        const long long j = (long long)q * d.Ng + g;
        float gl = 0.f;
        if (!CND(VC_SITE_DNU)) {
          const float x = b.lat[VC_SITE_DNU][j], sd = vel ? 0.01f : b.sd_dnu[j];
          float lik;
          if (vc_walk_by_batch(d)) {      // the partials of this batch's waves, in wave order (as the tails' gene blocks)
            int w0, nw;
            vc_walk_waves_of(d, q, &w0, &nw);
            lik = 0.f;
            for (int w = w0; w < w0 + nw; ++w) lik += sm[w][0][lane];
          } else {
            lik = vc_dnu_range_sum(d, b, g, q);
          }
          gl = -(lik - rw * x / (sd * sd));
        }
        G[d.poff[VC_P_DNU_LOCS] + j] = gl;
      }
    } else if (role == 12 && nb) {
      // ---- shape_inv: r-only terms of sum_c NB(k; r, eta): nmat*Nc*r*log r + sum_hist cnt*(lgamma(r+k)-lgamma(r))
      const float r = in[0];
      const float U_r = (d.kind == VC_KIND_PHASE) ? T(K) : (d.kind == VC_KIND_VFULL ? T(K + 2) : 0.f);
      const double lr = (double)logf(r);
      if (d.nmat_r > 0) loss -= (double)d.nmat_r * d.Nc * (double)r * lr + HLg;
      float gu = 0.f;
      if (!CND(VC_SITE_SHAPE_INV)) {
        const float si = in[1];
        const double dr = (double)U_r + (double)d.nmat_r * d.Nc * (lr + 1.0) + HDg;
        const double gsi = -(double)r * (double)r * dr + (double)rw * ((d.gamma_alpha - 1.f) / si - d.gamma_beta);
        gu = (float)(-gsi * (double)si);
      }
      G[d.poff[VC_P_SHAPE_INV_ULOCS] + g] = gu;
    } else if ((role == 13 || role == 14) && vel) {
      // ---- log gamma / log beta ----
      const float gam = in[0];
      float U_lb, U_lg;
      if (d.kind == VC_KIND_VFULL) { U_lb = -T(K); U_lg = T(K + 1) * gam; }
      else { U_lb = -T(0); U_lg = T(1) * gam; }
      float g_lg = 0.f, g_lb = 0.f;   // total d log p / d site (0 when the site is conditioned)
      if (!CND(VC_SITE_LOGGAMMA)) g_lg = U_lg - rw * (in[1] - in[3]) / (in[2] * in[2]);
      if (!CND(VC_SITE_LOGBETA)) g_lb = U_lb - rw * (in[4] - in[6]) / (in[5] * in[5]);
      if (!lrmn) {
        if (role == 13) {
          const float eg = in[7], eb = in[8];
          const bool cg = CND(VC_SITE_LOGGAMMA), cb = CND(VC_SITE_LOGBETA);
          G[d.poff[VC_P_LOGGAMMA_LOCS] + g] = -g_lg;
          G[d.poff[VC_P_LOGGAMMA_USCALES] + g] = cg ? 0.f : -g_lg * expf(in[9]) * eg - rw;
          G[d.poff[VC_P_LOGBETA_LOCS] + g] = -g_lb;
          G[d.poff[VC_P_LOGBETA_USCALES] + g] = cb ? 0.f : -g_lb * expf(in[10]) * eb - rw;
        }
      } else {
        // q(log beta | log gamma) = N(a + rho s_b delta / s_gamma, s_b sqrt(1-rho^2)); log gamma = loc + delta
        const bool cb = CND(VC_SITE_LOGBETA);
        const float A = g_lb;
        const float ent = cb ? 0.f : rw;        // weight of the guide's -log(std) term
        const float delta = in[7], sgam = in[8];
        const float sb = expf(in[9]);
        const float rho_real = in[10];
        const float sg = sigmoidf_(rho_real / d.rho_scale);
        const float rho = sg * 1.998f - 0.999f;
        const float om = 1.f - rho * rho, sq = sqrtf(om);
        const float dl_ddelta = -g_lg - A * rho * sb / sgam;
        const float dl_dsg = A * rho * sb * delta / (sgam * sgam);
        if (role == 13) {
          const float eb = in[11];
          G[d.poff[VC_P_LOGBETA_LOCS] + g] = -A;
          G[d.poff[VC_P_LOGBETA_USCALES] + g] = -A * (rho * delta / sgam + sq * eb) * sb - ent;
          float g_rho = -A * (sb * delta / sgam - sb * rho * eb / sq) + ent * rho / om;
          float g_rr = g_rho * 1.998f * sg * (1.f - sg) / d.rho_scale;
          if (!CND(VC_SITE_RHO_REAL)) g_rr += rw * (rho_real - d.rho_mean) / (d.rho_std * d.rho_std);
          G[d.poff[VC_P_RHO_REAL_LOC] + g] = g_rr;
          G[d.poff[VC_P_LRMN_LOC] + g] = -g_lg;
          const float dg = expf(in[12]);
          const float ed = in[13];
          G[d.poff[VC_P_LRMN_UCOV_DIAG] + g] = (dl_ddelta * ed / (2.f * sqrtf(dg)) + dl_dsg / (2.f * sgam)) * dg;
        } else {
#pragma unroll
          for (int k = 0; k < VC_MAX_RANK; ++k)
            if (k < d.R) {
              const long long j = d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)g * d.R + k;
              const float w = expf(in[11 + k]);
              const float ew = b.eps_used[d.eoff[VC_E_LRMN_W] + k];          // wave-uniform, a scalar-cache hit
              G[j] = (w > 0.f) ? (dl_ddelta * ew + dl_dsg * w / sgam) * w : 0.f;
            }
        }
      }
    }
  }
  VC_KSTAMP(1, 2);
  // only the shape_inv role (wave 12) carries loss terms: its wave sum is the block's partial, no block reduction
  if (role == 12) {
    const double tot = vc_wave_sum_d(loss);
    if (lane == 0) b.LP[d.nb_pre_gene + d.nb_pre_cell + gblock] = tot;
  }
}

// ---------------------------------------------------------------------------------------------
// K_post_cell
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void vc_post_cell_block(const VcDims& d, const VcBufs& b, float* __restrict__ G,
                                                   int cblock) {
  __shared__ float sm_w[16][VC_MAX_NW];
  const int c = cblock * 1024 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  float A[3] = {0.f, 0.f, 0.f};
  float2 sc1 = make_float2(0.f, 1.f);    // sin phi, cos phi of the cell (from its record)
  float dx01[2] = {0.f, 0.f};             // design-matrix entries of the first two conditions
  if (c < d.Nc) {
    // everything this cell needs is requested in one go (the partial sums and the site values are independent loads)
    const bool need_xy = d.poff[VC_P_PHIXY_LOCS] >= 0 && !CND(VC_SITE_PHIXY);
    float2 xy = make_float2(1.f, 0.f), pxy = make_float2(0.f, 0.f);
    float om = 0.f, dom = 0.f;
    if (vel) {
      dx01[0] = b.Dm[c];
      if (d.Nx > 1) dx01[1] = b.Dm[(size_t)d.Nc + c];
    }
    if (need_xy) {
      xy = *reinterpret_cast<const float2*>(b.lat[VC_SITE_PHIXY] + 2 * (size_t)c);
      pxy = *reinterpret_cast<const float2*>(b.pxy + 2 * (size_t)c);
      if (d.kind == VC_KIND_VFULL) { om = b.lat_omega[c]; dom = b.lat_domega[c]; }
    }
    const int cp = vc_pos(b, c);
    { const float* ct = b.CT + (size_t)cp * d.ctw; sc1 = make_float2(ct[0], ct[2]); }     // {sin, sin}, {cos, cos} of the first harmonic
    for (int gb = 0; gb < d.nGB; ++gb)
      for (int j = 0; j < d.nco; ++j) A[j] += b.CO[((size_t)gb * d.nco + j) * d.Nc + cp];
    VC_KSTAMP(1, 1);
    if (d.poff[VC_P_PHIXY_LOCS] >= 0) {
      float gx = 0.f, gy = 0.f;
      if (!CND(VC_SITE_PHIXY)) {
        float dphi = A[0];
        if (d.kind == VC_KIND_VFULL) dphi += om * A[1] + A[2] * dom;
        const float x = xy.x, y = xy.y;
        const float inv = 1.0f / (x * x + y * y);
        gx = -(dphi * (-y * inv) - (x - pxy.x));
        gy = -(dphi * (x * inv) - (y - pxy.y));
      }
      G[d.poff[VC_P_PHIXY_LOCS] + 2LL * c] = gx;
      G[d.poff[VC_P_PHIXY_LOCS] + 2LL * c + 1] = gy;
    }
  }
  if (vel) {
    // partial sums of d loglik / d nu_omega[x,h] = sum_c A3_c D[x,c] zeta_omega_h(phi_c)
    const float a3 = (c < d.Nc) ? (d.kind == VC_KIND_VFULL ? A[2] : A[0]) : 0.f;
    float s1 = sc1.x, c1 = sc1.y;                 // (the cell record's own sin / cos: the bits K_pre built the basis from)
    float sk[VC_MAXH], ck[VC_MAXH];
    sk[0] = s1; ck[0] = c1;
    for (int k = 1; k < d.Hw && k < VC_MAXH; ++k) {
      sk[k] = sk[k - 1] * c1 + ck[k - 1] * s1;
      ck[k] = ck[k - 1] * c1 - sk[k - 1] * s1;
    }
    for (int xq = 0; xq < d.Nx; ++xq) {
      const float dx = (c < d.Nc) ? (xq < 2 ? dx01[xq] : b.Dm[(size_t)xq * d.Nc + c]) : 0.f;
      for (int h = 0; h < d.Nhw; ++h) {
        const float z = (h == 0) ? 1.f : ((h & 1) ? sk[(h - 1) >> 1] : ck[(h - 1) >> 1]);
        const float t = vc_wave_sum(a3 * dx * z);
        if (lane == 0) sm_w[wave][xq * d.Nhw + h] = t;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < d.NW) {
      const int j = threadIdx.x;
      float t = 0.f;
      for (int w = 0; w < 16; ++w) t += sm_w[w][j];
      b.PW[(size_t)cblock * d.NW + j] = t;
    }
  }
}

// one launch for both second-stage reductions: blocks [0, nb_post_gene) gene level, the rest cell level
template <int MQ, bool MULTI = false, int SPEC = 0>
__global__ __launch_bounds__(1024) void vc_post_kernel(const VcDims d, const VcBufs b0, const float* __restrict__ P,
                                                       float* __restrict__ G0, long long* __restrict__ step_dev,
                                                       const VcBufs* __restrict__ bs, const VcParticleGrads pg) {
  vc_spec_assume<SPEC>(d);
  // every reader of this step's counter (K_pre) has finished and nothing in this launch reads it:
  // advance it here, so that K_fin / the optimiser (which only read it) see step + 1 = the 1-based Adam step
  // (MULTI: blockIdx.y = particle, its workspaces bs[blockIdx.y], its gradient buffer pg.g[blockIdx.y]; step_dev is null)
  VcBufs bm;
  if (MULTI) bm = bs[blockIdx.y];
  const VcBufs& b = MULTI ? bm : b0;
  float* __restrict__ G = MULTI ? pg.g[blockIdx.y] : G0;
  VC_KSTAMP(1, 0);
  if (blockIdx.x == 0 && threadIdx.x == 0 && step_dev) *step_dev += 1;
  if ((int)blockIdx.x < d.nb_post_gene) vc_post_gene_block<MQ>(d, b, P, G, blockIdx.x);
  else vc_post_cell_block(d, b, G, blockIdx.x - d.nb_post_gene);
  VC_KSTAMP(1, 3);
}

// ---------------------------------------------------------------------------------------------
// K_fin: one block.  Loss in fp64 (deterministic order), angular-speed gradients, header.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void vc_fin_block(const VcDims& d, const VcBufs& b, const float* P, float* G,
                                             double* loss_dev, long long loss_slots, long long step) {
  // 1 + NW independent reductions (the loss; per angular-speed coefficient sum_c d loglik/d omega_c * D * zeta_omega).
  // Every load of the phase is issued before anything is summed: the loss partials are spread over all 256 threads,
  // coefficient j over the lanes of wave j % 4; fixed order -> deterministic; a single barrier ends the phase.
  __shared__ double sm_lossw[4];
  __shared__ float sm_up[VC_MAX_NW];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int nlp = d.nb_pre_gene + d.nb_pre_cell + d.nb_post_gene;
  // inputs of this thread's angular-speed output element (used after the reductions): x, sd, mu, parameter, eps
  float fin_in[5] = {0.f, 1.f, 0.f, 0.f, 0.f};
  const int fin_per = (d.model == VC_MODEL_VELOCITY && d.guide == VC_GUIDE_LRMN) ? d.R + 2 : 2, fin_tt = t;
  if (d.model == VC_MODEL_VELOCITY && fin_tt < d.NW * fin_per) {
    const int j = fin_tt / fin_per, c = fin_tt % fin_per;
    if (!CND(VC_SITE_NUOMEGA)) { fin_in[0] = b.lat[VC_SITE_NUOMEGA][j]; fin_in[1] = b.sd_w[j]; fin_in[2] = b.mu_w[j]; }
    if (d.guide != VC_GUIDE_LRMN) {
      if (c == 1) { fin_in[3] = P[d.poff[VC_P_NUOMEGA_USCALES] + j]; fin_in[4] = b.eps_used[d.eoff[VC_E_NUOMEGA] + j]; }
    } else {
      const long long i = (long long)d.Ng + j;
      if (c >= 1 && c <= d.R) { fin_in[3] = P[d.poff[VC_P_LRMN_UCOV_FACTOR] + i * d.R + (c - 1)]; fin_in[4] = b.eps_used[d.eoff[VC_E_LRMN_W] + (c - 1)]; }
      else if (c == d.R + 1) { fin_in[3] = P[d.poff[VC_P_LRMN_UCOV_DIAG] + i]; fin_in[4] = b.eps_used[d.eoff[VC_E_LRMN_D] + i]; }
    }
  }
  {
    double s = 0.0;
#pragma unroll 4
    for (int i = t; i < nlp; i += 256) s += b.LP[i];
#pragma unroll 4
    for (int i = t; i < d.n_main_wg; i += 256) s -= (double)b.LO[i];   // loss = -loglik
    double u[2] = {0.0, 0.0};              // up to two coefficients per wave without a second round trip
    const int nw = d.model == VC_MODEL_VELOCITY ? d.NW : 0;
    // the partials of d loglik / d nu_omega: K_post's cell blocks', or (pw_inline) K_main's workgroups'
    const float* __restrict__ PWs = d.pw_inline ? b.PWM : b.PW;
    const int n_pw = d.pw_inline ? d.n_main_wg : d.nb_post_cell, pw_ld = d.pw_inline ? d.pw_inline : d.NW;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = wv + 4 * q;
      if (j < nw)
        u[q] = vc_col_sum_d(PWs, n_pw, pw_ld, j, lane, u[q]);
    }
    s = vc_wave_sum_d(s);
    if (lane == 0) sm_lossw[wv] = s;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = wv + 4 * q;
      if (j < nw) { const double r = vc_wave_sum_d(u[q]); if (lane == 0) sm_up[j] = (float)r; }
    }
    for (int j = wv + 8; j < nw; j += 4) {          // more than 8 coefficients: the rest one after the other
      double r = 0.0;
      r = vc_col_sum_d(PWs, n_pw, pw_ld, j, lane, r);
      r = vc_wave_sum_d(r);
      if (lane == 0) sm_up[j] = (float)r;
    }
  }
  __syncthreads();
  if (t == 0) {
    const double loss = ((sm_lossw[0] + sm_lossw[1]) + (sm_lossw[2] + sm_lossw[3])) + b.const_loss;
    if (loss_dev) loss_dev[loss_slots > 1 ? (step % loss_slots) : 0] = loss;
    if (!isfinite(loss)) {                 // failure detection: latch the first step whose loss is NaN / Inf
      b.status[0] += 1;
      if (b.status[1] == 0) b.status[1] = step + 1;
    }
    const float hi = (float)loss;
    G[0] = hi;
    G[1] = (float)(loss - (double)hi);
    G[2] = 0.f;
    G[3] = 0.f;
  }
  // angular-speed coefficient j: d/d(nu_omega_j) = up_j + prior term; one thread per OUTPUT element (its inputs were
  // requested ahead of the reductions, see fin_in above)
  if (d.model == VC_MODEL_VELOCITY && fin_tt < d.NW * fin_per) {
    const bool lrmn = d.guide == VC_GUIDE_LRMN;
    const int j = fin_tt / fin_per, c = fin_tt % fin_per;
    const bool cnd = CND(VC_SITE_NUOMEGA);
    float gx = 0.f;
    if (!cnd) gx = sm_up[j] - d.root_w * (fin_in[0] - fin_in[2]) / (fin_in[1] * fin_in[1]);
    if (!lrmn) {
      if (c == 0) G[d.poff[VC_P_NUOMEGA_LOCS] + j] = -gx;
      else G[d.poff[VC_P_NUOMEGA_USCALES] + j] = cnd ? 0.f : -gx * expf(fin_in[3]) * fin_in[4] - d.root_w;
    } else {
      const long long i = (long long)d.Ng + j;
      if (c == 0) G[d.poff[VC_P_LRMN_LOC] + i] = -gx;
      else if (c <= d.R) {
        const long long q = d.poff[VC_P_LRMN_UCOV_FACTOR] + i * d.R + (c - 1);
        const float w = expf(fin_in[3]);
        G[q] = (w > 0.f) ? -gx * fin_in[4] * w : 0.f;
      } else {
        const float dg = expf(fin_in[3]);
        G[d.poff[VC_P_LRMN_UCOV_DIAG] + i] = -gx * fin_in[4] / (2.f * sqrtf(dg)) * dg;
      }
    }
  }
  // more than 256 output elements (NW * (R + 2) > 256): the rest the plain way
  if (d.model == VC_MODEL_VELOCITY) {
    const bool lrmn = d.guide == VC_GUIDE_LRMN;
    for (int tt = t + 256; tt < d.NW * fin_per; tt += 256) {
      const int j = tt / fin_per, c = tt % fin_per;
      const bool cnd = CND(VC_SITE_NUOMEGA);
      float gx = 0.f;
      if (!cnd) {
        const float x = b.lat[VC_SITE_NUOMEGA][j], sd = b.sd_w[j];
        gx = sm_up[j] - d.root_w * (x - b.mu_w[j]) / (sd * sd);
      }
      const long long i = (long long)d.Ng + j;       // only LRMN reaches here (2 NW <= 128 otherwise)
      if (!lrmn) continue;
      if (c == 0) G[d.poff[VC_P_LRMN_LOC] + i] = -gx;
      else if (c <= d.R) {
        const long long q = d.poff[VC_P_LRMN_UCOV_FACTOR] + i * d.R + (c - 1);
        const float w = expf(P[q]);
        G[q] = (w > 0.f) ? -gx * b.eps_used[d.eoff[VC_E_LRMN_W] + (c - 1)] * w : 0.f;
      } else {
        const float dg = expf(P[d.poff[VC_P_LRMN_UCOV_DIAG] + i]);
        G[d.poff[VC_P_LRMN_UCOV_DIAG] + i] = -gx * b.eps_used[d.eoff[VC_E_LRMN_D] + i] / (2.f * sqrtf(dg)) * dg;
      }
    }
  }
}

__global__ __launch_bounds__(256) void vc_fin_kernel(const VcDims d, const VcBufs b, const float* P, float* G,
                                                     double* loss_dev, long long loss_slots, long long step_host,
                                                     const long long* step_dev) {
  vc_fin_block(d, b, P, G, loss_dev, loss_slots, step_dev ? *step_dev - 1 : step_host);
}

// ---------------------------------------------------------------------------------------------
// ClippedAdam on the flat parameter buffer (pyro.optim.ClippedAdam restated, one launch):
//   lr_t = lr0 * lrd^t ; g = clamp(g, +-clip) ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//   p -= lr_t * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)            (t = 1-based step)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vc_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      long long n, double lr0, double lrd /* log */, double b1, double b2,
                                                      double b1l, double b2l, float eps, float clip, float wd, int kind,
                                                      const unsigned char* __restrict__ frozen, long long frozen_off, long long t_host,
                                                      const long long* __restrict__ t_dev,
                                                      const float* __restrict__ loss_hdr,
                                                      double* __restrict__ loss_ring, long long loss_slots) {
  __shared__ float s_step, s_c2;
  // multi-rank path: the all-reduced loss (float hi + lo in the gradient header) goes into the loss ring here
  if (loss_ring && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long t1 = t_dev ? *t_dev : t_host;
    loss_ring[loss_slots > 1 ? ((t1 - 1) % loss_slots) : 0] = (double)loss_hdr[0] + (double)loss_hdr[1];
  }
  if (threadIdx.x == 0) {      // lrd^t, b^t as exp(t log .) once per block (the logs come from the host)
    const long long t1 = t_dev ? *t_dev : t_host;
    s_step = vc_adam_step_size(t1, lr0, lrd, b1l, b2l, kind);   // lrd, b1l, b2l: logs
    s_c2 = vc_adam_c2(t1, b2l, kind);
  }
  __syncthreads();
  const float step_size = s_step, c2 = s_c2;
  const float fb1 = (float)b1, fb2 = (float)b2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float mi = m[i], vi = v[i];
    const float pn = vc_adam_elem(p[i], g[i], mi, vi, step_size, fb1, fb2, eps, clip, c2, vc_wd_at(wd, frozen, frozen_off + i));
    m[i] = mi;
    v[i] = vi;
    p[i] = pn;
  }
}

// K_fin + ClippedAdam in one launch (single rank).  Block 0 finishes the loss and the angular-speed gradients
// and then updates exactly those parameters; every block updates the rest, whose gradients K_post completed.
__global__ __launch_bounds__(256) void vc_fin_adam_kernel(const VcDims d, const VcBufs b, float* P, float* G,
                                                          double* loss_dev, long long loss_slots, long long step_host,
                                                          const long long* step_dev, float* __restrict__ m,
                                                          float* __restrict__ v, double lr0, double lrd /* log */,
                                                          double b1, double b2, double b1l, double b2l, float eps, float clip,
                                                          float wd, int kind, const unsigned char* __restrict__ frozen, int header, long long total) {
  __shared__ float s_step, s_c2;
  const long long t1 = step_dev ? *step_dev : step_host + 1;        // 1-based optimiser step
  if (threadIdx.x == 0) {
    s_step = vc_adam_step_size(t1, lr0, lrd, b1l, b2l, kind);   // lrd, b1l, b2l: logs
    s_c2 = vc_adam_c2(t1, b2l, kind);
  }
  // parameters whose gradient K_fin produces: nu_omega (mean-field) or the LRMN tail rows
  long long lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (d.model == VC_MODEL_VELOCITY) {
    if (d.guide != VC_GUIDE_LRMN) {
      lo[0] = d.poff[VC_P_NUOMEGA_LOCS]; hi[0] = lo[0] + d.NW;
      lo[1] = d.poff[VC_P_NUOMEGA_USCALES]; hi[1] = lo[1] + d.NW;
    } else {
      lo[0] = d.poff[VC_P_LRMN_LOC] + d.Ng; hi[0] = d.poff[VC_P_LRMN_LOC] + d.M;
      lo[1] = d.poff[VC_P_LRMN_UCOV_DIAG] + d.Ng; hi[1] = d.poff[VC_P_LRMN_UCOV_DIAG] + d.M;
      lo[2] = d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)d.Ng * d.R; hi[2] = d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)d.M * d.R;
    }
  }
  VC_KSTAMP(2, 0);
  if (blockIdx.x == 0) vc_fin_block(d, b, P, G, loss_dev, loss_slots, t1 - 1);
  __syncthreads();
  VC_KSTAMP(2, 1);
  const float step_size = s_step, c2 = s_c2;
  const float fb1 = (float)b1, fb2 = (float)b2;
  auto upd = [&](long long idx) {
    const long long j = idx - header;
    float mi = m[j], vi = v[j];
    const float pn = vc_adam_elem(P[idx], G[idx], mi, vi, step_size, fb1, fb2, eps, clip, c2, vc_wd_at(wd, frozen, idx));
    m[j] = mi;
    v[j] = vi;
    P[idx] = pn;
  };
  for (long long idx = header + (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const bool tail = (idx >= lo[0] && idx < hi[0]) || (idx >= lo[1] && idx < hi[1]) || (idx >= lo[2] && idx < hi[2]);
    if (!tail) upd(idx);
  }
  VC_KSTAMP(2, 2);
  if (blockIdx.x == 0)
    for (int q = 0; q < 3; ++q)
      for (long long idx = lo[q] + threadIdx.x; idx < hi[q]; idx += 256) upd(idx);
  VC_KSTAMP(2, 3);
}

// K-particle step, last launch: block 0 finishes every particle's loss and angular-speed gradients (K_fin of particle k on its
// workspaces and gradient buffer, one after the other), all blocks then average the K gradients in particle order --
// (g_0 + g_1 + ...) * (1 / K), what a host loop's sum and PyTorch's division by a scalar give; left in particle 0's buffer --
// and apply ClippedAdam to the average; block 0 files the averaged loss and sets the step counter to t + 1.
template <int SPEC = 0>
__global__ __launch_bounds__(256) void vc_particle_fin_adam_kernel(const VcDims d, const VcBufs* __restrict__ bs, const VcParticleGrads pg,
                                                                   float* P, double* loss_dev, long long loss_slots, long long step_host,
                                                                   long long* step_dev, double* __restrict__ scratch, float* __restrict__ m,
                                                                   float* __restrict__ v, double lr0, double lrd /* log */, double b1,
                                                                   double b2, double b1l, double b2l, float eps, float clip, float wd, int kind,
                                                                   const unsigned char* __restrict__ frozen, int header, long long total) {
  vc_spec_assume<SPEC>(d);
  __shared__ float s_step, s_c2;
  const int K = pg.K;
  const long long t1 = step_host + 1;        // 1-based optimiser step (the host's mirror of the device counter: nothing here reads it)
  if (threadIdx.x == 0) {
    s_step = vc_adam_step_size(t1, lr0, lrd, b1l, b2l, kind);   // lrd, b1l, b2l: logs
    s_c2 = vc_adam_c2(t1, b2l, kind);
  }
  long long lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};      // parameters whose gradient K_fin produces (vc_fin_adam_kernel)
  if (d.model == VC_MODEL_VELOCITY) {
    if (d.guide != VC_GUIDE_LRMN) {
      lo[0] = d.poff[VC_P_NUOMEGA_LOCS]; hi[0] = lo[0] + d.NW;
      lo[1] = d.poff[VC_P_NUOMEGA_USCALES]; hi[1] = lo[1] + d.NW;
    } else {
      lo[0] = d.poff[VC_P_LRMN_LOC] + d.Ng; hi[0] = d.poff[VC_P_LRMN_LOC] + d.M;
      lo[1] = d.poff[VC_P_LRMN_UCOV_DIAG] + d.Ng; hi[1] = d.poff[VC_P_LRMN_UCOV_DIAG] + d.M;
      lo[2] = d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)d.Ng * d.R; hi[2] = d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)d.M * d.R;
    }
  }
  if (blockIdx.x == 0) {
    for (int k = 0; k < K; ++k) {
      const VcBufs bk = bs[k];
      vc_fin_block(d, bk, P, pg.g[k], scratch + k, 1, step_host);
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      double l = 0.0;
      for (int k = 0; k < K; ++k) l += (double)pg.g[k][0] + (double)pg.g[k][1];
      const double avg = l * (1.0 / (double)K);
      const float h = (float)avg, lw = (float)(avg - (double)h);
      pg.g[0][0] = h; pg.g[0][1] = lw;
      if (loss_dev) loss_dev[loss_slots > 1 ? (step_host % loss_slots) : 0] = (double)h + (double)lw;
      if (step_dev) step_dev[0] = t1;
    }
  }
  __syncthreads();
  const float step_size = s_step, c2 = s_c2;
  const float fb1 = (float)b1, fb2 = (float)b2, inv = 1.0f / (float)K;
  auto upd = [&](long long idx) {
    const long long j = idx - header;
    float a = 0.f + pg.g[0][idx];
    for (int k = 1; k < K; ++k) a += pg.g[k][idx];
    const float ga = a * inv;
    pg.g[0][idx] = ga;
    float mi = m[j], vi = v[j];
    const float pn = vc_adam_elem(P[idx], ga, mi, vi, step_size, fb1, fb2, eps, clip, c2, vc_wd_at(wd, frozen, idx));
    m[j] = mi;
    v[j] = vi;
    P[idx] = pn;
  };
  for (long long idx = header + (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const bool tail = (idx >= lo[0] && idx < hi[0]) || (idx >= lo[1] && idx < hi[1]) || (idx >= lo[2] && idx < hi[2]);
    if (!tail) upd(idx);
  }
  if (blockIdx.x == 0)
    for (int q = 0; q < 3; ++q)
      for (long long idx = lo[q] + threadIdx.x; idx < hi[q]; idx += 256) upd(idx);
}
void vc_launch_particle_fin_adam(const VcDims& d, const VcBufs* bs_dev, const VcParticleGrads& pg, float* params, double* loss_dev,
                                 long long loss_slots, long long step, long long* step_dev, double* scratch, float* m, float* v,
                                 const VcAdamHyper& h, int header, long long total, hipStream_t st) {
  long long nb = (total - header + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  if (vc_spec_launch<VC_SPECK_PARTICLES, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_particle_fin_adam_kernel<decltype(sp)::value>), dim3((unsigned)nb), dim3(256), 0, st, d, bs_dev, pg, params, loss_dev,
                           loss_slots, step, step_dev, scratch, m, v, h.lr0, log(h.lrd), h.b1, h.b2, log(h.b1), log(h.b2), h.eps, h.clip, h.wd,
                           h.kind, h.frozen, header, total);
      }))
    return;
  hipLaunchKernelGGL((vc_particle_fin_adam_kernel<0>), dim3((unsigned)nb), dim3(256), 0, st, d, bs_dev, pg, params, loss_dev, loss_slots, step,
                     step_dev, scratch, m, v, h.lr0, log(h.lrd), h.b1, h.b2, log(h.b1), log(h.b2), h.eps, h.clip, h.wd, h.kind, h.frozen, header, total);
}

void vc_launch_fin_adam(const VcDims& d, const VcBufs& b, float* params, float* grad, double* loss_dev,
                        long long loss_slots, long long step, long long* step_dev, float* m, float* v, const VcAdamHyper& h,
                        int header, long long total, hipStream_t st) {
  long long nb = (total - header + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(vc_fin_adam_kernel, dim3((unsigned)nb), dim3(256), 0, st, d, b, params, grad, loss_dev, loss_slots,
                     step, step_dev, m, v, h.lr0, log(h.lrd), h.b1, h.b2, log(h.b1), log(h.b2), h.eps, h.clip, h.wd, h.kind, h.frozen, header, total);
}

void vc_launch_adam(float* p, const float* g, float* m, float* v, long long n, const VcAdamHyper& h, long long t_host,
                    const long long* t_dev, const float* loss_hdr, double* loss_ring, long long loss_slots, hipStream_t st) {
  long long nb = (n + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(vc_adam_kernel, dim3((unsigned)nb), dim3(256), 0, st, p, g, m, v, n, h.lr0, log(h.lrd), h.b1, h.b2, log(h.b1),
                     log(h.b2), h.eps, h.clip, h.wd, h.kind, h.frozen, h.frozen_off, t_host, t_dev, loss_hdr, loss_ring, loss_slots);
}

void vc_launch_post(const VcDims& d, const VcBufs& b, const float* params, float* grad, long long* step_dev,
                    hipStream_t st) {
  if (d.generic) { vc_launch_post_generic(d, b, params, grad, step_dev, st); return; }
  const dim3 grid(d.nb_post_gene + d.nb_post_cell), block(1024);
  if (d.nq <= 2) hipLaunchKernelGGL((vc_post_kernel<2, false>), grid, block, 0, st, d, b, params, grad, step_dev, (const VcBufs*)nullptr, VcParticleGrads());
  else if (d.nq <= 6) hipLaunchKernelGGL((vc_post_kernel<6, false>), grid, block, 0, st, d, b, params, grad, step_dev, (const VcBufs*)nullptr, VcParticleGrads());
  else hipLaunchKernelGGL((vc_post_kernel<VC_MAXQ, false>), grid, block, 0, st, d, b, params, grad, step_dev, (const VcBufs*)nullptr, VcParticleGrads());
}
void vc_launch_post_particles(const VcDims& d, const VcBufs& b, const VcBufs* bs_dev, const VcParticleGrads& pg, const float* params,
                              hipStream_t st) {
  const dim3 grid(d.nb_post_gene + d.nb_post_cell, pg.K), block(1024);
  long long* no_ctr = nullptr;
  float* no_g = nullptr;
  if (vc_spec_launch<VC_SPECK_PARTICLES, 0>(d.spec, [&](auto mq, auto sp) {
        constexpr int MQP = decltype(mq)::value <= 2 ? 2 : (decltype(mq)::value <= 6 ? 6 : VC_MAXQ);      // (K_post's row bounds: 2 | 6 | all)
        hipLaunchKernelGGL((vc_post_kernel<MQP, true, decltype(sp)::value>), grid, block, 0, st, d, b, params, no_g, no_ctr, bs_dev, pg);
      }))
    return;
  if (d.nq <= 2) hipLaunchKernelGGL((vc_post_kernel<2, true>), grid, block, 0, st, d, b, params, no_g, no_ctr, bs_dev, pg);
  else if (d.nq <= 6) hipLaunchKernelGGL((vc_post_kernel<6, true>), grid, block, 0, st, d, b, params, no_g, no_ctr, bs_dev, pg);
  else hipLaunchKernelGGL((vc_post_kernel<VC_MAXQ, true>), grid, block, 0, st, d, b, params, no_g, no_ctr, bs_dev, pg);
}

void vc_launch_fin(const VcDims& d, const VcBufs& b, const float* params, float* grad, double* loss_dev,
                   long long loss_slots, long long step, const long long* step_dev, hipStream_t st) {
  if (d.generic) { vc_launch_fin_generic(d, b, params, grad, loss_dev, loss_slots, step, step_dev, st); return; }
  hipLaunchKernelGGL(vc_fin_kernel, dim3(1), dim3(256), 0, st, d, b, params, grad, loss_dev, loss_slots, step,
                     step_dev);
}

// ---------------------------------------------------------------------------------------------
// E[log S] / E[log U] summaries of posterior_sampling (velocity_inference_model.py:236-258): one thread per 4
// consecutive cells of a gene row, float4 stores (write-bound: 4 outputs of Ng*Nc floats).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vc_expected_logs_kernel(const VcDims d, const float* __restrict__ cf,
                                                               const float* __restrict__ Dbm,
                                                               const float* __restrict__ nu, const float* __restrict__ dnu,
                                                               const float* __restrict__ phi, const float* __restrict__ omega,
                                                               const float* __restrict__ logbeta, const float* __restrict__ gamma,
                                                               float cf_avg, float* __restrict__ oS, float* __restrict__ oS2,
                                                               float* __restrict__ oU, float* __restrict__ oU2) {
  const int g = blockIdx.y;
  const long long c0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c0 >= d.Nc) return;
  const float* __restrict__ nug = nu + (size_t)g * d.Nh;          // (any number of harmonics: read where they are)
  const float lb = logbeta ? logbeta[g] : 0.f, gm = gamma ? gamma[g] : 0.f;
  float s[4], s2[4], u[4], u2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long c = c0 + j < d.Nc ? c0 + j : d.Nc - 1;
    const float p = phi[c];
    float es = nug[0], dd = 0.f;
    for (int k = 1; k <= d.H; ++k) {
      float sn, cs;
      sincosf((float)k * p, &sn, &cs);
      es += nug[2 * k - 1] * sn + nug[2 * k] * cs;
      dd += (float)k * (nug[2 * k - 1] * cs - nug[2 * k] * sn);
    }
    for (int q = 0; q < d.Nb; ++q) es += Dbm[(size_t)q * d.Nc + c] * dnu[(size_t)q * d.Ng + g];
    s[j] = es + cf[c];
    s2[j] = es + cf_avg;
    if (oU) {
      const float core = -lb + logf(fmaxf(dd * omega[c] + gm, 0.f) + 1e-5f);
      u[j] = core + s[j];
      u2[j] = core + s2[j];
    }
  }
  const size_t o = (size_t)g * d.Nc + c0;
  if (c0 + 3 < d.Nc && (d.Nc & 3) == 0) {
    *reinterpret_cast<float4*>(oS + o) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(oS2 + o) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    if (oU) {
      *reinterpret_cast<float4*>(oU + o) = make_float4(u[0], u[1], u[2], u[3]);
      *reinterpret_cast<float4*>(oU2 + o) = make_float4(u2[0], u2[1], u2[2], u2[3]);
    }
  } else {
    for (int j = 0; j < 4 && c0 + j < d.Nc; ++j) {
      oS[o + j] = s[j]; oS2[o + j] = s2[j];
      if (oU) { oU[o + j] = u[j]; oU2[o + j] = u2[j]; }
    }
  }
}

void vc_launch_expected_logs(const VcDims& d, const VcBufs& b, const float* nu, const float* dnu, const float* phi,
                             const float* omega, const float* logbeta, const float* gamma, float cf_avg, float* out_S,
                             float* out_S2, float* out_U, float* out_U2, hipStream_t st) {
  const unsigned nbx = (unsigned)((d.Nc + 1023) / 1024);
  hipLaunchKernelGGL(vc_expected_logs_kernel, dim3(nbx, d.Ng), dim3(256), 0, st, d, b.cf, b.Dbm, nu, dnu, phi, omega,
                     logbeta, gamma, cf_avg, out_S, out_S2, out_U, out_U2);
}
