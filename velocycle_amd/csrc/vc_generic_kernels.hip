// The run-time-sized kernel set: the same SVI step for configurations OUTSIDE the compiled fast set -- more than 3 harmonics
// (reference utils.py:400-437 takes any n_harmonics), more than 4 batches (preprocessing.py:65-93 builds one design column per
// unique id; phase_inference_model.py:374-377, velocity_inference_model.py:360), LRMN rank > 8 (preprocessing.py:239,
// velocity_inference_guide.py:91-92), more than 64 angular-speed coefficients.  Nothing here is a template over H / Nb / rank:
// harmonics, batches and low-rank columns are loops, the per-gene state of the likelihood kernel lives in the LDS instead of
// registers.  Slower than the fast set by design (the fast instantiations stay selected wherever they exist); the arithmetic is
// the fast kernels' statement by statement, so both are held against the same float64 oracle at the same tolerances
// (tests/test_hip_sweep.py).  Only the unfused sequence exists here: K_pre -> K_main -> K_post -> K_fin -> ClippedAdam
// (vc_elbo_grad + vc_clipped_adam; cells sharded: the all-reduce of the gradient buffer between them).
//
//   vc_main_generic_kernel<KIND, NOISE>   one wave per workgroup: lane = 2 genes (one packed pair), wave = one 128-gene block x a
//                                         run of cells; nu~[k] and d loglik / d nu~[k] of the lane's pair in the LDS, [k][lane]
//                                         (conflict-free 8-byte accesses), K KB per wave; cell records through scalar loads
//   vc_pre_generic_kernel                 thread = gene / thread = cell, every site one after the other
//   vc_post_generic_kernel                thread = gene (each needed row of K_main's partials summed over the chunks) / thread = cell
//   vc_fin_generic_kernel                 loss assembly + the nu_omega / LRMN tail gradients for any number of coefficients
#pragma clang fp contract(off)
#include "vc_main_math.h"        // v2f helpers, observation models

// ---------------------------------------------------------------------------------------------
// K_main, generic
// ---------------------------------------------------------------------------------------------
template <int KIND, int NOISE>
__global__ __launch_bounds__(64) void vc_main_generic_kernel(const VcDims d, const VcBufs b) {
  constexpr bool HAS_S = (KIND != VC_KIND_VU);
  constexpr bool HAS_U = (KIND != VC_KIND_PHASE);
  constexpr bool FULL = (KIND == VC_KIND_VFULL);
  constexpr bool LN = (NOISE == VC_NOISE_LOGNORMAL);
  constexpr bool L2 = VC_FOLD_LOG2E && !LN;
  constexpr bool HLB = VC_HOIST_LB && FULL && !LN;
  constexpr int NCO = FULL ? 3 : 1;
  constexpr float CO_SCALE = L2 ? VC_LN2 : 1.f;
  static_assert(VC_OMEGA_CS && VC_NR_MERGE, "the generic kernel restates the default arithmetic of the fast one");
  extern __shared__ v2f lds_g[];                 // [2 K][64]: nu~ rows, then the gradient accumulators
  const int lane = threadIdx.x;
  const int H = d.H, NH = 2 * H + 1, NB = d.with_dnu ? d.Nb : 0, K = d.K;
  const int gb = blockIdx.x % d.nGB, chunk = blockIdx.x / d.nGB;
  const int g0 = gb * 128 + 2 * lane;
  const size_t NP = d.Ng_pad;
  v2f* nu = lds_g + lane;                        // nu[k * 64]
  v2f* gnu = lds_g + (size_t)K * 64 + lane;      // gnu[k * 64]
  int my_cw;
  long long cbeg;
  {
    typedef const __attribute__((address_space(4))) int* ciptr;
    ciptr tl = (ciptr)(const void*)(b.wg_tile + 4 * (size_t)blockIdx.x);
    my_cw = tl[1];
    cbeg = (long long)tl[0];
  }
  long long cend = cbeg + my_cw;
  if (cbeg > d.Nc) cbeg = d.Nc;
  if (cend > d.Nc) cend = d.Nc;
  const int ncell = (int)(cend - cbeg);
  // per-gene latents
  const float* gt = b.GT + g0;
  const float2 lbv = *reinterpret_cast<const float2*>(gt + (size_t)K * NP);
  const float2 gmv = *reinterpret_cast<const float2*>(gt + (size_t)(K + 1) * NP);
  const float2 rrv = *reinterpret_cast<const float2*>(gt + (size_t)(K + 2) * NP);
  for (int k = 0; k < K; ++k) {
    const float2 v = *reinterpret_cast<const float2*>(gt + (size_t)k * NP);
    v2f x = v2f{v.x, v.y};
    if (k == 0 && KIND == VC_KIND_VU) x -= v2f{lbv.x, lbv.y};         // -log beta folded into the constant harmonic
    if (L2) x *= VC_LOG2E;
    nu[(size_t)k * 64] = x;
    gnu[(size_t)k * 64] = v2(0.f);
  }
  const v2f lb2 = HLB ? v2(0.f) : v2f{lbv.x, lbv.y} * VC_LOG2E;
  const v2f ib = v2f{__expf(-lbv.x), __expf(-lbv.y)};
  const v2f gam = v2f{gmv.x, gmv.y}, rr = v2f{rrv.x, rrv.y};
  const float inv_s2_s = 1.0f / (d.sigma_ln_s * d.sigma_ln_s);
  const float inv_s2_u = 1.0f / (d.sigma_ln_u * d.sigma_ln_u);
  v2f gau = v2(0.f), gw = v2(0.f), ll = v2(0.f), lt = v2(0.f);
  const float* Sp = HAS_S ? b.S + ((size_t)gb * d.Nc) * 128 + 2 * lane : nullptr;
  const float* Up = HAS_U ? b.U + ((size_t)gb * d.Nc) * 128 + 2 * lane : nullptr;
  const int r_db = 2 * H, r_om = 2 * H + NB, r_cf = r_om + 1, r_x = r_om + 2;      // record layout, in {x, x} pairs

  for (int i = 0; i < ncell; ++i) {
    const long long c = cbeg + i;
    typedef const __attribute__((address_space(4))) v2f* cptr;                     // wave-uniform record: scalar loads
    cptr rec = (cptr)(const void*)(b.CT + (size_t)c * d.ctw);
    v2f sv = v2(0.f), uv = v2(0.f);
    if (HAS_S) { const float2 t = *reinterpret_cast<const float2*>(Sp + (size_t)c * 128); sv = v2f{t.x, t.y}; }
    if (HAS_U) { const float2 t = *reinterpret_cast<const float2*>(Up + (size_t)c * 128); uv = v2f{t.x, t.y}; }
    // eta_S = nu . zeta(phi) + Db . dnu + cf ;  dd = nu . zeta'(phi) ;  e2 = nu . zeta''(phi)
    v2f es = nu[0] + rec[r_cf];
    v2f dd = v2(0.f), e2 = v2(0.f);
    for (int k = 0; k < H; ++k) {
      const v2f ns = nu[(size_t)(2 * k + 1) * 64], nc = nu[(size_t)(2 * k + 2) * 64];
      const v2f sn = rec[2 * k], cs = rec[2 * k + 1];
      const float kk = (float)(k + 1);
      if (FULL) {
        const v2f t = v2_fma(ns, sn, nc * cs);
        es += t;
        e2 = v2_fma(t, v2(-(kk * kk)), e2);
      } else {
        es = v2_fma(ns, sn, v2_fma(nc, cs, es));
      }
      const v2f u = v2_fma(ns, cs, -(nc * sn));
      dd = v2_fma(u, v2(kk), dd);
    }
    for (int q = 0; q < NB; ++q) es = v2_fma(nu[(size_t)(NH + q) * 64], rec[r_db + q], es);
    const v2f es2 = L2 ? es : es * VC_LOG2E;
    v2f a = v2(0.f), w = v2(0.f), muS = v2(0.f);
    if (HAS_S) {
      v2f aS;
      if (LN) vc_obs_lognormal(sv, es, inv_s2_s, aS, ll);
      else {
        muS = v2_exp2(es2);
        vc_obs_counts<NOISE>(sv, es2, muS, rr, aS, ll, lt);
      }
      a += aS;
    }
    if (HAS_U) {
      const v2f z = v2_fma(dd, rec[r_om], gam);
      v2f m;
      asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(m) : "v"(z), "v"(v2(1.2676506e30f)));
      const v2f zp = v2_fma(z, m, v2(1e-5f));
      const v2f eu2 = (KIND == VC_KIND_VU || HLB) ? es2 + v2_log2(zp) : (es2 - lb2) + v2_log2(zp);
      v2f aU;
      if (NOISE == VC_NOISE_NB) {
        const v2f muU = FULL ? muS * (ib * zp) : v2_exp2(eu2);
        const v2f t = rr + muU;
        const v2f lt2 = v2_log2(t);
        const v2f R = v2_rcp(t * zp);
        const v2f nR = (rr * (uv - muU)) * R;
        aU = nR * zp;
        w = nR * m;
        ll = v2_fma(uv, eu2 - lt2, ll);
        lt += lt2;
      } else {
        const v2f q = v2_rcp(zp) * m;
        if (LN) vc_obs_lognormal(uv, eu2 * VC_LN2, inv_s2_u, aU, ll);
        else {
          const v2f muU = FULL ? muS * (ib * zp) : v2_exp2(eu2);
          vc_obs_counts<NOISE>(uv, eu2, muU, rr, aU, ll, lt);
        }
        w = aU * q;
      }
      a += aU;
      gau += aU;
      gw += w;
    }
    v2f A1 = v2(0.f), A2 = v2(0.f), A3 = v2(0.f);
    if (KIND != VC_KIND_VU) {
      gnu[0] += a;
      for (int k = 0; k < H; ++k) {
        const v2f sn = rec[2 * k], cs = rec[2 * k + 1];
        v2f g1 = gnu[(size_t)(2 * k + 1) * 64], g2 = gnu[(size_t)(2 * k + 2) * 64];
        if (FULL) {          // d z / d nu_s = k omega cos k phi, d z / d nu_c = -k omega sin k phi: both in the record
          g1 = v2_fma(a, sn, v2_fma(w, rec[r_x + 2 * k], g1));
          g2 = v2_fma(a, cs, v2_fma(-w, rec[r_x + 2 * k + 1], g2));
        } else {
          g1 = v2_fma(a, sn, g1);
          g2 = v2_fma(a, cs, g2);
        }
        gnu[(size_t)(2 * k + 1) * 64] = g1;
        gnu[(size_t)(2 * k + 2) * 64] = g2;
      }
      for (int q = 0; q < NB; ++q) gnu[(size_t)(NH + q) * 64] = v2_fma(a, rec[r_db + q], gnu[(size_t)(NH + q) * 64]);
      A1 = a * dd;
    }
    if (FULL) A2 = w * e2;
    if (HAS_U) A3 = w * dd;
    // per-cell sums over the genes of this wave
    float p0, p1 = 0.f, p2 = 0.f;
    if (KIND == VC_KIND_PHASE) p0 = A1.x + A1.y;
    else if (KIND == VC_KIND_VU) p0 = A3.x + A3.y;
    else { p0 = A1.x + A1.y; p1 = A2.x + A2.y; p2 = A3.x + A3.y; }
    const float t0 = vc_wave_sum(p0);
    float t1 = 0.f, t2 = 0.f;
    if (NCO == 3) { t1 = vc_wave_sum(p1); t2 = vc_wave_sum(p2); }
    if (lane == 0) {
      float* co = b.CO + ((size_t)gb * NCO) * d.Nc + c;
      co[0] = t0 * CO_SCALE;
      if (NCO == 3) { co[(size_t)d.Nc] = t1 * CO_SCALE; co[2 * (size_t)d.Nc] = t2 * CO_SCALE; }
    }
  }
  // ---- epilogue: one row of partials per workgroup (= per wave) -------------------------------
  const float nobs = (float)ncell * (FULL ? 2.f : 1.f);
  const v2f g0v = gnu[0];
  const v2f gr = (NOISE == VC_NOISE_NB && KIND != VC_KIND_VU) ? lt * (-VC_LN2) - nobs - g0v * v2_rcp(rr) : v2(0.f);
  {
    if (HLB && chunk == 0) {      // sum_c k_U * (-log2 beta) over ALL of this rank's cells, once per gene
      const float2 su = *reinterpret_cast<const float2*>(b.gene_sum_u + g0);
      ll -= v2f{lbv.x * su.x, lbv.y * su.y} * VC_LOG2E;
    }
    const v2f lj = ((NOISE == VC_NOISE_NB) ? ll - rr * lt : ll) * VC_LN2;
    float l = 0.f;
    l += (g0 < d.Ng) ? lj.x : 0.f;
    l += (g0 + 1 < d.Ng) ? lj.y : 0.f;
    l = vc_wave_sum(l);
    if (lane == 0) b.LO[blockIdx.x] = l;
  }
  const int NQ = d.nq;
  float* go = b.GO + ((size_t)chunk * NQ) * NP + g0;
  auto put = [&](int q, v2f v) { *reinterpret_cast<float2*>(go + (size_t)q * NP) = make_float2(v.x, v.y); };
  if (KIND == VC_KIND_VU) { put(0, gau); put(1, gw); }
  else {
    for (int k = 0; k < K; ++k) put(k, gnu[(size_t)k * 64]);
    if (KIND == VC_KIND_PHASE) put(K, gr);
    else { put(K, gau); put(K + 1, gw); put(K + 2, gr); }
  }
}

template <int KIND, int NOISE>
static void vc_main_generic_launch(const VcDims& d, const VcBufs& b, hipStream_t st) {
  const unsigned dyn = (unsigned)(2 * d.K * 64 * sizeof(v2f));
  hipLaunchKernelGGL((vc_main_generic_kernel<KIND, NOISE>), dim3(d.n_main_wg), dim3(64), dyn, st, d, b);
}

vc_main_launch_fn vc_find_generic_main_kernel(int kind, int noise, const void** kernel) {
#define VC_GEN_CASE(KIND, NOISE)                                                         \
  if (kind == KIND && noise == NOISE) {                                                  \
    if (kernel) *kernel = (const void*)&vc_main_generic_kernel<KIND, NOISE>;             \
    return &vc_main_generic_launch<KIND, NOISE>;                                         \
  }
  VC_GEN_CASE(VC_KIND_PHASE, VC_NOISE_NB) VC_GEN_CASE(VC_KIND_PHASE, VC_NOISE_POISSON) VC_GEN_CASE(VC_KIND_PHASE, VC_NOISE_LOGNORMAL)
  VC_GEN_CASE(VC_KIND_VFULL, VC_NOISE_NB) VC_GEN_CASE(VC_KIND_VFULL, VC_NOISE_POISSON) VC_GEN_CASE(VC_KIND_VFULL, VC_NOISE_LOGNORMAL)
  VC_GEN_CASE(VC_KIND_VU, VC_NOISE_NB) VC_GEN_CASE(VC_KIND_VU, VC_NOISE_POISSON) VC_GEN_CASE(VC_KIND_VU, VC_NOISE_LOGNORMAL)
#undef VC_GEN_CASE
  return nullptr;
}

// ---------------------------------------------------------------------------------------------
// K_pre, generic: blocks [0, nb_pre_gene) 256 genes each, [.., + nb_pre_cell) 256 cells each, then the histogram blocks
// (statements of vc_pre_kernel; reference: velocity_inference_guide.py:9-141, phase_inference_guide.py:10-56, the priors of
// velocity_inference_model.py:322-353,383 / phase_inference_model.py:360-366,392)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vc_pre_generic_kernel(const VcDims d, const VcBufs b, const float* __restrict__ P,
                                                             const float* __restrict__ eps_in, uint64_t seed, long long step_host,
                                                             const long long* __restrict__ step_dev, int cond_only, int particles,
                                                             int particle) {
  __shared__ double sm_red[16];
  extern __shared__ float s_nuw[];               // [NW] the nu_omega sample of this step (cell blocks)
  const long long step = (step_dev ? *step_dev : step_host) * particles + particle;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  const bool nb = d.noise == VC_NOISE_NB;
  const size_t NP = d.Ng_pad;
  const int K = d.K;
  double loss = 0.0;
  auto eps = [&](long long li, long long gi) { return vc_eps(eps_in, b.eps_used, seed, step, li, gi); };
  if ((int)blockIdx.x >= d.nb_pre_gene + d.nb_pre_cell) {
    const int hb = blockIdx.x - d.nb_pre_gene - d.nb_pre_cell;
    if (d.hist_dense) {          // (the histogram blocks use the launch's dynamic LDS for themselves: no s_nuw there)
      vc_hist_dense_quarter(d, b, hb >> 2, hb & 3, P, cond_only, 0, vc_hist_lds());
      return;
    }
    const int task = hb * 4 + (threadIdx.x >> 6);
    if (task < b.n_tasks) vc_hist_wave(d, b, P, cond_only, task, threadIdx.x & 63);
    return;
  }
  if ((int)blockIdx.x < d.nb_pre_gene) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < d.Ng_pad) {
      float* GT = b.GT + g;
      if (g >= d.Ng) {          // padded gene: nu~ = 0 (never reaches a per-cell sum), loss masked in K_main
        for (int k = 0; k < K; ++k) GT[k * NP] = 0.f;
        GT[K * NP] = 0.f; GT[(K + 1) * NP] = 1.f; GT[(K + 2) * NP] = 1.f;
      } else {
        float logp = 0.f, logq = 0.f;
        for (int h = 0; h < d.Nh; ++h) {                                     // ---- nu[h]
          const long long j = (long long)g * d.Nh + h;
          float x;
          if (cond_only) x = CND(VC_SITE_NU) ? b.cnd[VC_SITE_NU][j] : 0.f;
          else {
            const float e = eps(d.eoff[VC_E_NU] + j, d.eoff[VC_E_NU] + j);
            const float u = P[d.poff[VC_P_NU_USCALES] + j];
            const float xg = P[d.poff[VC_P_NU_LOCS] + j] + expf(u) * e;
            if (CND(VC_SITE_NU)) x = b.cnd[VC_SITE_NU][j];
            else { x = xg; logq += -0.5f * e * e - u - 0.5f * VC_LOG_2PI; }
            logp += vc_normal_lp(x, b.mu_nu[j], b.sd_nu[j]);
            b.lat[VC_SITE_NU][j] = x;
          }
          GT[h * NP] = x;
        }
        for (int q = 0; q < d.Nb && d.with_dnu; ++q) {                       // ---- delta nu (Delta guide)
          const long long j = (long long)q * d.Ng + g;
          float x;
          if (cond_only) x = CND(VC_SITE_DNU) ? b.cnd[VC_SITE_DNU][j] : 0.f;
          else {
            x = CND(VC_SITE_DNU) ? b.cnd[VC_SITE_DNU][j] : P[d.poff[VC_P_DNU_LOCS] + j];
            logp += vc_normal_lp(x, 0.f, vel ? 0.01f : b.sd_dnu[j]);
            b.lat[VC_SITE_DNU][j] = x;
          }
          GT[(d.Nh + q) * NP] = x;
        }
        {                                                                    // ---- shape_inv (Delta guide, positive)
          float si = 1.f;
          if (nb) {
            if (cond_only) si = CND(VC_SITE_SHAPE_INV) ? b.cnd[VC_SITE_SHAPE_INV][g] : 1.f;
            else {
              si = CND(VC_SITE_SHAPE_INV) ? b.cnd[VC_SITE_SHAPE_INV][g] : expf(P[d.poff[VC_P_SHAPE_INV_ULOCS] + g]);
              logp += d.gamma_alpha * logf(d.gamma_beta) + (d.gamma_alpha - 1.f) * logf(si) - d.gamma_beta * si - d.lgamma_alpha;
              b.lat[VC_SITE_SHAPE_INV][g] = si;
            }
          }
          GT[(K + 2) * NP] = 1.0f / si;
        }
        float lg = 0.f, lbv = 0.f;                                           // ---- log gamma, log beta
        if (vel && !cond_only) {
          float lg_guide, lb_guide;
          if (!lrmn) {
            const float eg = eps(d.eoff[VC_E_LOGGAMMA] + g, d.eoff[VC_E_LOGGAMMA] + g);
            const float eb = eps(d.eoff[VC_E_LOGBETA] + g, d.eoff[VC_E_LOGBETA] + g);
            const float ug = P[d.poff[VC_P_LOGGAMMA_USCALES] + g], ub = P[d.poff[VC_P_LOGBETA_USCALES] + g];
            lg_guide = P[d.poff[VC_P_LOGGAMMA_LOCS] + g] + expf(ug) * eg;
            lb_guide = P[d.poff[VC_P_LOGBETA_LOCS] + g] + expf(ub) * eb;
            if (!CND(VC_SITE_LOGGAMMA)) logq += -0.5f * eg * eg - ug - 0.5f * VC_LOG_2PI;
            if (!CND(VC_SITE_LOGBETA)) logq += -0.5f * eb * eb - ub - 0.5f * VC_LOG_2PI;
          } else {
            // LowRankMultivariateNormal.rsample: X = loc + W eps_W + sqrt(cov_diag) eps_D, any rank
            float delta = 0.f, w2 = 0.f;
            for (int k = 0; k < d.R; ++k) {
              const float w = expf(P[d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)g * d.R + k]);
              const float ew = eps(d.eoff[VC_E_LRMN_W] + k, d.eoff[VC_E_LRMN_W] + k);
              delta += w * ew;
              w2 += w * w;
            }
            const float dg = expf(P[d.poff[VC_P_LRMN_UCOV_DIAG] + g]);
            const float ed = eps(d.eoff[VC_E_LRMN_D] + g, d.eoff[VC_E_LRMN_D] + g);
            delta += sqrtf(dg) * ed;
            const float sgam = sqrtf(w2 + dg);
            lg_guide = P[d.poff[VC_P_LRMN_LOC] + g] + delta;
            const float rho_real_g = P[d.poff[VC_P_RHO_REAL_LOC] + g];
            const float rho = sigmoidf_(rho_real_g / d.rho_scale) * 1.998f - 0.999f;
            const float ub = P[d.poff[VC_P_LOGBETA_USCALES] + g];
            const float sb = expf(ub);
            const float eb = eps(d.eoff[VC_E_LOGBETA] + g, d.eoff[VC_E_LOGBETA] + g);
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
        GT[K * NP] = lbv;
        GT[(K + 1) * NP] = expf(lg);
        loss = -(double)d.root_w * ((double)logp - (double)logq);
      }
    }
  } else {
    // ------------------------------- cell part ------------------------------------------------
    const int bc = blockIdx.x - d.nb_pre_gene;
    if (vel) {
      for (int j = threadIdx.x; j < d.NW; j += 256) {
        float val = 0.f, lq = 0.f;
        if (!cond_only) {
          if (!lrmn) {
            const float e = eps(d.eoff[VC_E_NUOMEGA] + j, d.eoff[VC_E_NUOMEGA] + j);
            const float u = P[d.poff[VC_P_NUOMEGA_USCALES] + j];
            val = P[d.poff[VC_P_NUOMEGA_LOCS] + j] + expf(u) * e;
            lq = -0.5f * e * e - u - 0.5f * VC_LOG_2PI;
          } else {
            const long long i = (long long)d.Ng + j;
            float delta = 0.f;
            for (int k = 0; k < d.R; ++k) {
              const float ew = eps_in ? eps_in[d.eoff[VC_E_LRMN_W] + k] : vc_philox_normal(seed, step, d.eoff[VC_E_LRMN_W] + k);
              delta += expf(P[d.poff[VC_P_LRMN_UCOV_FACTOR] + i * d.R + k]) * ew;
            }
            const float ed = eps(d.eoff[VC_E_LRMN_D] + i, d.eoff[VC_E_LRMN_D] + i);
            delta += sqrtf(expf(P[d.poff[VC_P_LRMN_UCOV_DIAG] + i])) * ed;
            val = P[d.poff[VC_P_LRMN_LOC] + i] + delta;
            if (bc == 0) b.lat_delta[i] = delta;
          }
          const float x = CND(VC_SITE_NUOMEGA) ? b.cnd[VC_SITE_NUOMEGA][j] : val;
          if (bc == 0) {
            b.lat[VC_SITE_NUOMEGA][j] = x;
            const float lp = vc_normal_lp(x, b.mu_w[j], b.sd_w[j]);
            loss += -(double)d.root_w * ((double)lp - ((CND(VC_SITE_NUOMEGA) || lrmn) ? 0.0 : (double)lq));
          }
          val = x;
        }
        s_nuw[j] = val;
      }
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
        const float ex = eps(li, gi), ey = eps(li + 1, gi + 1);
        const float px = b.pxy[2 * c], py = b.pxy[2 * c + 1];
        if (CND(VC_SITE_PHIXY)) {
          x = b.cnd[VC_SITE_PHIXY][2 * c]; y = b.cnd[VC_SITE_PHIXY][2 * c + 1];
          loss += 0.5 * ((double)(x - px) * (x - px) + (double)(y - py) * (y - py)) + (double)VC_LOG_2PI;
        } else {
          x = P[d.poff[VC_P_PHIXY_LOCS] + 2LL * c] + ex;
          y = P[d.poff[VC_P_PHIXY_LOCS] + 2LL * c + 1] + ey;
          loss += 0.5 * ((double)(x - px) * (x - px) + (double)(y - py) * (y - py)) - 0.5 * ((double)ex * ex + (double)ey * ey);
        }
        b.lat[VC_SITE_PHIXY][2 * c] = x;
        b.lat[VC_SITE_PHIXY][2 * c + 1] = y;
      }
      const float phi = atan2f(y, x);                      // utils.py:505-506 (the deterministic site)
      float s1, c1;
      vc_dir_sincos(x, y, &s1, &c1);
      float2* ct = reinterpret_cast<float2*>(b.CT + (size_t)c * d.ctw);
      const int nbk = d.with_dnu ? d.Nb : 0;
      const int hm = d.H > d.Hw ? d.H : d.Hw;
      // harmonics by the angle-addition recurrence (as the fast kernels build them); omega = sum_x D[x,c] (nu_w[x,0] + sum_k ...)
      float omega = 0.f, domega = 0.f;
      if (vel && !cond_only)
        for (int xq = 0; xq < d.Nx; ++xq) omega += b.Dm[(size_t)xq * d.Nc + c] * s_nuw[xq * d.Nhw];
      float sk = s1, ck = c1;
      for (int k = 0; k < hm; ++k) {
        if (k > 0) { const float sn = sk * c1 + ck * s1, cn = ck * c1 - sk * s1; sk = sn; ck = cn; }
        if (k < d.H) { ct[2 * k] = make_float2(sk, sk); ct[2 * k + 1] = make_float2(ck, ck); }
        if (vel && !cond_only && k < d.Hw)
          for (int xq = 0; xq < d.Nx; ++xq) {
            const float dx = b.Dm[(size_t)xq * d.Nc + c];
            const float ws = s_nuw[xq * d.Nhw + 2 * k + 1], wc = s_nuw[xq * d.Nhw + 2 * k + 2];
            omega += dx * (ws * sk + wc * ck);
            domega += dx * ((float)(k + 1) * (ws * ck - wc * sk));
          }
      }
      for (int q = 0; q < nbk; ++q) { const float v = b.Dbm[(size_t)q * d.Nc + c]; ct[2 * d.H + q] = make_float2(v, v); }
      { const float oz = omega * vc_rec_omega_scale(d.noise); ct[2 * d.H + nbk] = make_float2(oz, oz); }
      { const float cfs = b.cf[c] * vc_rec_cf_scale(d.noise); ct[2 * d.H + nbk + 1] = make_float2(cfs, cfs); }
      if (d.kind == VC_KIND_VFULL) {          // k omega cos k phi, k omega sin k phi (read by the S+U kernel)
        sk = s1; ck = c1;
        for (int k = 0; k < d.H; ++k) {
          if (k > 0) { const float sn = sk * c1 + ck * s1, cn = ck * c1 - sk * s1; sk = sn; ck = cn; }
          const float wk = (float)(k + 1) * omega;
          ct[2 * d.H + nbk + 2 + 2 * k] = make_float2(wk * ck, wk * ck);
          ct[2 * d.H + nbk + 3 + 2 * k] = make_float2(wk * sk, wk * sk);
        }
      }
      b.lat_phi[c] = phi;
      b.lat_omega[c] = omega;
      b.lat_domega[c] = domega;
    }
  }
  const double tot = vc_block_sum_d(loss, sm_red);
  if (threadIdx.x == 0) b.LP[blockIdx.x] = cond_only ? 0.0 : tot;
}

void vc_launch_pre_generic(const VcDims& d, const VcBufs& b, const float* params, const float* eps, uint64_t seed, long long step,
                           const long long* step_dev, int cond_only, int with_hist, hipStream_t st, int particles, int particle) {
  const int nb_hist = with_hist ? vc_hist_blocks_pre(d, b) : 0;
  unsigned dyn = (unsigned)(sizeof(float) * (d.NW > 0 ? d.NW : 1));
  if (vc_hist_dyn_lds(d, with_hist, 256) > dyn) dyn = vc_hist_dyn_lds(d, with_hist, 256);
  hipLaunchKernelGGL(vc_pre_generic_kernel, dim3(d.nb_pre_gene + d.nb_pre_cell + nb_hist), dim3(256), dyn, st, d, b, params, eps,
                     seed, step, step_dev, cond_only, particles, particle);
}

// ---------------------------------------------------------------------------------------------
// K_post, generic (statements of vc_post_gene_block / vc_post_cell_block)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vc_post_generic_kernel(const VcDims d, const VcBufs b, const float* __restrict__ P,
                                                              float* __restrict__ G, long long* __restrict__ step_dev) {
  __shared__ double sm_red[16];
  __shared__ float sm_w4[4];
  if (blockIdx.x == 0 && threadIdx.x == 0 && step_dev) *step_dev += 1;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  const bool nb = d.noise == VC_NOISE_NB;
  const int K = d.K, Nh = d.Nh;
  const float rw = d.root_w;
  const size_t NP = d.Ng_pad;
  if ((int)blockIdx.x < d.nb_post_gene) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    double loss = 0.0;
    if (g < d.Ng) {
      // row q of K_main's partials summed over the chunks, eight requested per trip
      auto T = [&](int q) {
        float acc = 0.f;
        constexpr int UB = 8;
        for (int ch0 = 0; ch0 < d.n_chunks; ch0 += UB) {
          float v[UB];
#pragma unroll
          for (int u = 0; u < UB; ++u) v[u] = b.GO[((size_t)(ch0 + u < d.n_chunks ? ch0 + u : ch0) * d.nq + q) * NP + g];
#pragma unroll
          for (int u = 0; u < UB; ++u) if (ch0 + u < d.n_chunks) acc += v[u];
        }
        return acc;
      };
      for (int h = 0; h < Nh; ++h) {
        const long long j = (long long)g * Nh + h;
        float gl = 0.f, gu = 0.f;
        if (!CND(VC_SITE_NU)) {
          const float x = b.lat[VC_SITE_NU][j], sd = b.sd_nu[j];
          const float gx = T(h) - rw * (x - b.mu_nu[j]) / (sd * sd);
          gl = -gx;
          gu = -gx * expf(P[d.poff[VC_P_NU_USCALES] + j]) * b.eps_used[d.eoff[VC_E_NU] + j] - rw;
        }
        G[d.poff[VC_P_NU_LOCS] + j] = gl;
        G[d.poff[VC_P_NU_USCALES] + j] = gu;
      }
      for (int q = 0; q < d.Nb && d.with_dnu; ++q) {
        const long long j = (long long)q * d.Ng + g;
        float gl = 0.f;
        if (!CND(VC_SITE_DNU)) {
          const float x = b.lat[VC_SITE_DNU][j], sd = vel ? 0.01f : b.sd_dnu[j];
          gl = -(T(Nh + q) - rw * x / (sd * sd));
        }
        G[d.poff[VC_P_DNU_LOCS] + j] = gl;
      }
      if (nb) {
        const float r = b.GT[(size_t)(K + 2) * NP + g];
        double HLg = 0.0, HDg = 0.0;
        for (int t = b.h_tptr[g]; t < b.h_tptr[g + 1]; ++t) { HLg += b.HL[t]; HDg += b.HD[t]; }
        const float U_r = (d.kind == VC_KIND_PHASE) ? T(K) : (d.kind == VC_KIND_VFULL ? T(K + 2) : 0.f);
        if (d.nmat_r > 0) loss -= (double)d.nmat_r * d.Nc * (double)r * (double)logf(r) + HLg;
        float gu = 0.f;
        if (!CND(VC_SITE_SHAPE_INV)) gu = vc_si_grad(d, r, b.lat[VC_SITE_SHAPE_INV][g], U_r, HDg, rw);
        G[d.poff[VC_P_SHAPE_INV_ULOCS] + g] = gu;
      }
      if (vel) {
        const float gam = b.GT[(size_t)(K + 1) * NP + g];
        float U_lb, U_lg;
        if (d.kind == VC_KIND_VFULL) { U_lb = -T(K); U_lg = T(K + 1) * gam; }
        else { U_lb = -T(0); U_lg = T(1) * gam; }
        float g_lg = 0.f, g_lb = 0.f;
        if (!CND(VC_SITE_LOGGAMMA)) { const float sd = b.sd_g[g]; g_lg = U_lg - rw * (b.lat[VC_SITE_LOGGAMMA][g] - b.mu_g[g]) / (sd * sd); }
        if (!CND(VC_SITE_LOGBETA)) { const float sd = b.sd_b[g]; g_lb = U_lb - rw * (b.lat[VC_SITE_LOGBETA][g] - b.mu_b[g]) / (sd * sd); }
        if (!lrmn) {
          const float eg = b.eps_used[d.eoff[VC_E_LOGGAMMA] + g], eb = b.eps_used[d.eoff[VC_E_LOGBETA] + g];
          const bool cg = CND(VC_SITE_LOGGAMMA), cb = CND(VC_SITE_LOGBETA);
          G[d.poff[VC_P_LOGGAMMA_LOCS] + g] = -g_lg;
          G[d.poff[VC_P_LOGGAMMA_USCALES] + g] = cg ? 0.f : -g_lg * expf(P[d.poff[VC_P_LOGGAMMA_USCALES] + g]) * eg - rw;
          G[d.poff[VC_P_LOGBETA_LOCS] + g] = -g_lb;
          G[d.poff[VC_P_LOGBETA_USCALES] + g] = cb ? 0.f : -g_lb * expf(P[d.poff[VC_P_LOGBETA_USCALES] + g]) * eb - rw;
        } else {
          const bool cb = CND(VC_SITE_LOGBETA);
          const float A = g_lb;
          const float ent = cb ? 0.f : rw;
          const float delta = b.lat_delta[g], sgam = b.lat_sgam[g];
          const float sb = expf(P[d.poff[VC_P_LOGBETA_USCALES] + g]);
          const float rho_real = P[d.poff[VC_P_RHO_REAL_LOC] + g];
          const float sg = sigmoidf_(rho_real / d.rho_scale);
          const float rho = sg * 1.998f - 0.999f;
          const float om = 1.f - rho * rho, sq = sqrtf(om);
          const float dl_ddelta = -g_lg - A * rho * sb / sgam;
          const float dl_dsg = A * rho * sb * delta / (sgam * sgam);
          const float eb = b.eps_used[d.eoff[VC_E_LOGBETA] + g];
          G[d.poff[VC_P_LOGBETA_LOCS] + g] = -A;
          G[d.poff[VC_P_LOGBETA_USCALES] + g] = -A * (rho * delta / sgam + sq * eb) * sb - ent;
          float g_rho = -A * (sb * delta / sgam - sb * rho * eb / sq) + ent * rho / om;
          float g_rr = g_rho * 1.998f * sg * (1.f - sg) / d.rho_scale;
          if (!CND(VC_SITE_RHO_REAL)) g_rr += rw * (rho_real - d.rho_mean) / (d.rho_std * d.rho_std);
          G[d.poff[VC_P_RHO_REAL_LOC] + g] = g_rr;
          G[d.poff[VC_P_LRMN_LOC] + g] = -g_lg;
          const float dg = expf(P[d.poff[VC_P_LRMN_UCOV_DIAG] + g]);
          const float ed = b.eps_used[d.eoff[VC_E_LRMN_D] + g];
          G[d.poff[VC_P_LRMN_UCOV_DIAG] + g] = (dl_ddelta * ed / (2.f * sqrtf(dg)) + dl_dsg / (2.f * sgam)) * dg;
          for (int k = 0; k < d.R; ++k) {
            const long long j = d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)g * d.R + k;
            const float w = expf(P[j]);
            const float ew = b.eps_used[d.eoff[VC_E_LRMN_W] + k];
            G[j] = (w > 0.f) ? (dl_ddelta * ew + dl_dsg * w / sgam) * w : 0.f;
          }
        }
      }
    }
    const double tot = vc_block_sum_d(loss, sm_red);
    if (threadIdx.x == 0) b.LP[d.nb_pre_gene + d.nb_pre_cell + blockIdx.x] = tot;
    return;
  }
  // ------------------------------- cell part ------------------------------------------------------
  const int cblock = blockIdx.x - d.nb_post_gene;
  const int c = cblock * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float A[3] = {0.f, 0.f, 0.f};
  float s1 = 0.f, c1 = 1.f;
  if (c < d.Nc) {
    { const float* ct = b.CT + (size_t)c * d.ctw; s1 = ct[0]; c1 = ct[2]; }     // the cell record's own sin / cos
    for (int gb = 0; gb < d.nGB; ++gb)
      for (int j = 0; j < d.nco; ++j) A[j] += b.CO[((size_t)gb * d.nco + j) * d.Nc + c];
    if (d.poff[VC_P_PHIXY_LOCS] >= 0) {
      float gx = 0.f, gy = 0.f;
      if (!CND(VC_SITE_PHIXY)) {
        float dphi = A[0];
        if (d.kind == VC_KIND_VFULL) dphi += b.lat_omega[c] * A[1] + A[2] * b.lat_domega[c];
        const float x = b.lat[VC_SITE_PHIXY][2 * c], y = b.lat[VC_SITE_PHIXY][2 * c + 1];
        const float inv = 1.0f / (x * x + y * y);
        gx = -(dphi * (-y * inv) - (x - b.pxy[2 * c]));
        gy = -(dphi * (x * inv) - (y - b.pxy[2 * c + 1]));
      }
      G[d.poff[VC_P_PHIXY_LOCS] + 2LL * c] = gx;
      G[d.poff[VC_P_PHIXY_LOCS] + 2LL * c + 1] = gy;
    }
  }
  if (vel) {
    // partial sums of d loglik / d nu_omega[x,h] = sum_c A3_c D[x,c] zeta_omega_h(phi_c), one coefficient after the other
    const float a3 = (c < d.Nc) ? (d.kind == VC_KIND_VFULL ? A[2] : A[0]) : 0.f;
    for (int xq = 0; xq < d.Nx; ++xq) {
      const float dx = (c < d.Nc) ? b.Dm[(size_t)xq * d.Nc + c] : 0.f;
      float sk = s1, ck = c1;
      for (int h = 0; h < d.Nhw; ++h) {
        if (h >= 3 && (h & 1)) { const float sn = sk * c1 + ck * s1, cn = ck * c1 - sk * s1; sk = sn; ck = cn; }
        const float z = (h == 0) ? 1.f : ((h & 1) ? sk : ck);
        const float t = vc_wave_sum(a3 * dx * z);
        __syncthreads();
        if (lane == 0) sm_w4[wave] = t;
        __syncthreads();
        if (threadIdx.x == 0) b.PW[(size_t)cblock * d.NW + xq * d.Nhw + h] = (sm_w4[0] + sm_w4[1]) + (sm_w4[2] + sm_w4[3]);
      }
    }
  }
}

void vc_launch_post_generic(const VcDims& d, const VcBufs& b, const float* params, float* grad, long long* step_dev, hipStream_t st) {
  hipLaunchKernelGGL(vc_post_generic_kernel, dim3(d.nb_post_gene + d.nb_post_cell), dim3(256), 0, st, d, b, params, grad, step_dev);
}

// ---------------------------------------------------------------------------------------------
// K_fin, generic: one block (statements of vc_fin_block)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vc_fin_generic_kernel(const VcDims d, const VcBufs b, const float* P, float* G, double* loss_dev,
                                                             long long loss_slots, long long step_host, const long long* step_dev) {
  __shared__ double sm_lossw[4];
  extern __shared__ float s_up[];               // [NW] sum over the cell blocks of the partials of d loglik / d nu_omega
  const long long step = step_dev ? *step_dev - 1 : step_host;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int nlp = d.nb_pre_gene + d.nb_pre_cell + d.nb_post_gene;
  double s = 0.0;
  for (int i = t; i < nlp; i += 256) s += b.LP[i];
  for (int i = t; i < d.n_main_wg; i += 256) s -= (double)b.LO[i];
  s = vc_wave_sum_d(s);
  if (lane == 0) sm_lossw[wv] = s;
  const int nw = d.model == VC_MODEL_VELOCITY ? d.NW : 0;
  for (int j = wv; j < nw; j += 4) {
    double r = 0.0;
    for (int i = lane; i < d.nb_post_cell; i += 64) r += (double)b.PW[(size_t)i * d.NW + j];
    r = vc_wave_sum_d(r);
    if (lane == 0) s_up[j] = (float)r;
  }
  __syncthreads();
  if (t == 0) {
    const double loss = ((sm_lossw[0] + sm_lossw[1]) + (sm_lossw[2] + sm_lossw[3])) + b.const_loss;
    if (loss_dev) loss_dev[loss_slots > 1 ? (step % loss_slots) : 0] = loss;
    if (!isfinite(loss)) {
      b.status[0] += 1;
      if (b.status[1] == 0) b.status[1] = step + 1;
    }
    const float hi = (float)loss;
    G[0] = hi;
    G[1] = (float)(loss - (double)hi);
    G[2] = 0.f;
    G[3] = 0.f;
  }
  if (d.model != VC_MODEL_VELOCITY) return;
  const bool lrmn = d.guide == VC_GUIDE_LRMN;
  const int fin_per = lrmn ? d.R + 2 : 2;
  const bool cnd = CND(VC_SITE_NUOMEGA);
  for (int tt = t; tt < d.NW * fin_per; tt += 256) {
    const int j = tt / fin_per, c = tt % fin_per;
    float gx = 0.f;
    if (!cnd) {
      const float x = b.lat[VC_SITE_NUOMEGA][j], sd = b.sd_w[j];
      gx = s_up[j] - d.root_w * (x - b.mu_w[j]) / (sd * sd);
    }
    if (!lrmn) {
      if (c == 0) G[d.poff[VC_P_NUOMEGA_LOCS] + j] = -gx;
      else G[d.poff[VC_P_NUOMEGA_USCALES] + j] = cnd ? 0.f : -gx * expf(P[d.poff[VC_P_NUOMEGA_USCALES] + j]) * b.eps_used[d.eoff[VC_E_NUOMEGA] + j] - d.root_w;
    } else {
      const long long i = (long long)d.Ng + j;
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

void vc_launch_fin_generic(const VcDims& d, const VcBufs& b, const float* params, float* grad, double* loss_dev, long long loss_slots,
                           long long step, const long long* step_dev, hipStream_t st) {
  const unsigned dyn = (unsigned)(sizeof(float) * (d.NW > 0 ? d.NW : 1));
  hipLaunchKernelGGL(vc_fin_generic_kernel, dim3(1), dim3(256), dyn, st, d, b, params, grad, loss_dev, loss_slots, step, step_dev);
}
