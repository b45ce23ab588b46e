// The likelihood + reparameterised-gradient kernel (K_main): one pass over the blocked count
// matrices, fusing the ~110 full-size ATen ops (forward + autograd backward) that the reference
// spends per SVI step in velocity_latent_variable_model (velocity_inference_model.py:344-386) /
// phase_latent_variable_model (phase_inference_model.py:369-395).
//
// Mapping (CDNA4): lane = 4 consecutive genes, wave = one 256-gene block x `cw` consecutive cells,
// workgroup = 4 waves on the same gene block (4*cw consecutive cells).
//   * per-gene sampled latents (nu~, log beta, gamma, r) and the per-gene gradient accumulators stay
//     in VGPRs for the whole cell loop -- no cross-lane traffic for gene-level sums;
//   * the per-cell record (sin k phi, cos k phi, Db[:,c], omega_c, cf_c) is wave-uniform: it is read
//     with scalar loads (s_load_dwordx8) and consumed as SGPR operands;
//   * counts are streamed with one global_load_dwordx4 per lane per matrix per cell: 1 KiB
//     contiguous per wave-instruction, each wave walking a private contiguous region of HBM
//     ([gene block][cell][256] layout);
//   * per-cell sums over genes (for d/dphi and d/domega) are 64-lane DPP reductions, staged one
//     lane per cell and flushed as a coalesced 256-B store every 64 cells;
//   * gene-level partials of the 4 waves are combined through LDS and written once per workgroup;
//     the second (deterministic) reduction stage is K_post.  No float atomics anywhere.
//
// Bound: HBM.  Algorithmic bytes: 4*Ng*Nc per matrix read (S and U for VFULL, one matrix otherwise).
#pragma once
#include "vc_common.h"

template <int H, int NB>
struct VcCellRec {
  float sn[H], cs[H];
  float db[NB > 0 ? NB : 1];
  float omega, cf;
};

template <int H, int NB>
__device__ __forceinline__ VcCellRec<H, NB> vc_load_cell(const float* __restrict__ ct) {
  VcCellRec<H, NB> r;
#pragma unroll
  for (int k = 0; k < H; ++k) { r.sn[k] = ct[2 * k]; r.cs[k] = ct[2 * k + 1]; }
#pragma unroll
  for (int b = 0; b < NB; ++b) r.db[b] = ct[2 * H + b];
  r.omega = ct[2 * H + NB];
  r.cf = ct[2 * H + NB + 1];
  return r;
}

// Observation model of one count k under log-mean eta.  eta2 = eta * log2(e): the hardware
// transcendentals are base 2 (v_exp_f32 / v_log_f32), so everything logarithmic is carried in log2
// units and rescaled once per gene after the cell loop.  Inputs of the raw instructions are never
// denormal here (t = r + mu >= r > 0), so no range fix-up code is needed around them.
//   a    = d loglik / d eta                       (natural units)
//   lacc += k * (eta2 - log2 t)   [NB]  | k*eta2 - mu*log2e [Poisson] | -0.5 e^2/s^2 [Lognormal]
//   tacc += log2 t                [NB]: sum_c log(r + mu) enters both the loss (times r) and d/dr;
// the remaining NB pieces need no per-element work: sum_c (r+k)/(r+mu) = Nc + (sum_c a)/r, and the
// r-only terms (r log r, lgamma) come from the per-gene count histograms in K_post.
#define VC_LOG2E 1.4426950408889634f
#define VC_LN2 0.6931471805599453f

template <int NOISE>
__device__ __forceinline__ void vc_obs(float k, float eta, float eta2, float r, float inv_s2, float& a,
                                       float& lacc, float& tacc) {
  if (NOISE == VC_NOISE_NB) {
    const float mu = __builtin_amdgcn_exp2f(eta2);
    const float t = r + mu;
    const float lt2 = __builtin_amdgcn_logf(t);
    const float it = __builtin_amdgcn_rcpf(t);
    a = (r * (k - mu)) * it;
    lacc = fmaf(k, eta2 - lt2, lacc);
    tacc += lt2;
  } else if (NOISE == VC_NOISE_POISSON) {
    const float mu = __builtin_amdgcn_exp2f(eta2);
    a = k - mu;
    lacc = fmaf(k, eta2, lacc) - mu * VC_LOG2E;
  } else {  // Lognormal: k already holds log(count + 1); lacc in natural units / LN2 to share the rescale
    const float e = k - eta;
    a = e * inv_s2;
    lacc = fmaf(-0.5f * VC_LOG2E * e, a, lacc);
  }
}

template <int H, int NB, int KIND, int NOISE>
__global__ __launch_bounds__(256) void vc_main_kernel(const VcDims d, const VcBufs b) {
  constexpr int NH = 2 * H + 1;
  constexpr int K = NH + NB;
  constexpr bool HAS_S = (KIND != VC_KIND_VU);
  constexpr bool HAS_U = (KIND != VC_KIND_PHASE);
  constexpr bool FULL = (KIND == VC_KIND_VFULL);
  constexpr int NQ = (KIND == VC_KIND_PHASE) ? K + 1 : (KIND == VC_KIND_VFULL ? K + 3 : 2);
  constexpr int NCO = FULL ? 3 : 1;

  const int lane = threadIdx.x & 63;
  // wave index as an SGPR value, so that everything derived from it (cell range, cell-record
  // addresses) is provably wave-uniform and the records are fetched with scalar loads
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gb = blockIdx.x % d.nGB;
  const int chunk = blockIdx.x / d.nGB;
  const int gl = lane * 4;                 // gene offset inside the block
  const int g0 = gb * VC_GBW + gl;

  // ---- per-gene latents into registers ----------------------------------------------------
  float nu[K][4], lb[4], gam[4], rr[4];
  {
    const float* gt = b.GT + g0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(gt + (size_t)k * d.Ng_pad);
      nu[k][0] = v.x; nu[k][1] = v.y; nu[k][2] = v.z; nu[k][3] = v.w;
    }
    const float4 v0 = *reinterpret_cast<const float4*>(gt + (size_t)K * d.Ng_pad);
    const float4 v1 = *reinterpret_cast<const float4*>(gt + (size_t)(K + 1) * d.Ng_pad);
    const float4 v2 = *reinterpret_cast<const float4*>(gt + (size_t)(K + 2) * d.Ng_pad);
    lb[0] = v0.x; lb[1] = v0.y; lb[2] = v0.z; lb[3] = v0.w;
    gam[0] = v1.x; gam[1] = v1.y; gam[2] = v1.z; gam[3] = v1.w;
    rr[0] = v2.x; rr[1] = v2.y; rr[2] = v2.z; rr[3] = v2.w;
  }
  const float inv_s2_s = 1.0f / (d.sigma_ln_s * d.sigma_ln_s);
  const float inv_s2_u = 1.0f / (d.sigma_ln_u * d.sigma_ln_u);
  float lb2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lb2[j] = lb[j] * VC_LOG2E;

  // ---- accumulators ---------------------------------------------------------------------------
  float gnu[K][4];     // d loglik / d nu~[k]
  float gau[4], gw[4];   // sum_c aU, sum_c aU * d etaU/dz
  float ll[4], lt[4];    // log2-unit accumulators: likelihood pieces, sum_c log2(r + mu)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int k = 0; k < K; ++k) gnu[k][j] = 0.f;
    gau[j] = gw[j] = ll[j] = lt[j] = 0.f;
  }

  long long cbeg = (long long)chunk * (VC_WAVES * d.cw) + (long long)wave * d.cw;
  long long cend = cbeg + d.cw;
  if (cbeg > d.Nc) cbeg = d.Nc;
  if (cend > d.Nc) cend = d.Nc;

  const size_t blk_base = ((size_t)gb * d.Nc) * VC_GBW + gl;
  const float* Sp = HAS_S ? b.S + blk_base : nullptr;
  const float* Up = HAS_U ? b.U + blk_base : nullptr;

  for (long long cb = cbeg; cb < cend; cb += 64) {
    const int n = (int)((cend - cb) < 64 ? (cend - cb) : 64);
    float keep0 = 0.f, keep1 = 0.f, keep2 = 0.f;

    // software pipeline: counts and cell record of the next cell are in flight while this one is processed
    float4 s_nx, u_nx;
    if (HAS_S) s_nx = *reinterpret_cast<const float4*>(Sp + (size_t)cb * VC_GBW);
    if (HAS_U) u_nx = *reinterpret_cast<const float4*>(Up + (size_t)cb * VC_GBW);
    VcCellRec<H, NB> rec_nx = vc_load_cell<H, NB>(b.CT + (size_t)cb * d.ctw);

    for (int i = 0; i < n; ++i) {
      const long long c = cb + i;
      float4 s4, u4;
      if (HAS_S) s4 = s_nx;
      if (HAS_U) u4 = u_nx;
      const VcCellRec<H, NB> rec = rec_nx;
      if (i + 1 < n) {
        if (HAS_S) s_nx = *reinterpret_cast<const float4*>(Sp + (size_t)(c + 1) * VC_GBW);
        if (HAS_U) u_nx = *reinterpret_cast<const float4*>(Up + (size_t)(c + 1) * VC_GBW);
        rec_nx = vc_load_cell<H, NB>(b.CT + (size_t)(c + 1) * d.ctw);
      }
      const float sv[4] = {HAS_S ? s4.x : 0.f, HAS_S ? s4.y : 0.f, HAS_S ? s4.z : 0.f, HAS_S ? s4.w : 0.f};
      const float uv[4] = {HAS_U ? u4.x : 0.f, HAS_U ? u4.y : 0.f, HAS_U ? u4.z : 0.f, HAS_U ? u4.w : 0.f};

      float A1 = 0.f, A2 = 0.f, A3 = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // eta_S = nu . zeta(phi) + Db . dnu + cf ;  dd = nu . zeta'(phi) ;  e2 = nu . zeta''(phi)
        float es = nu[0][j] + rec.cf;
        float dd = 0.f, e2 = 0.f;
#pragma unroll
        for (int k = 0; k < H; ++k) {
          const float ns = nu[2 * k + 1][j], nc = nu[2 * k + 2][j];
          const float t = ns * rec.sn[k] + nc * rec.cs[k];
          es += t;
          dd += (float)(k + 1) * (ns * rec.cs[k] - nc * rec.sn[k]);
          e2 -= (float)((k + 1) * (k + 1)) * t;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) es += nu[NH + q][j] * rec.db[q];

        const float es2 = es * VC_LOG2E;
        float a = 0.f, w = 0.f;
        if (HAS_S) {
          float aS;
          vc_obs<NOISE>(sv[j], es, es2, rr[j], inv_s2_s, aS, ll[j], lt[j]);
          a += aS;
        }
        if (HAS_U) {
          // eta_U = -log beta + log(relu(dd * omega + gamma) + 1e-5) + eta_S
          const float z = fmaf(dd, rec.omega, gam[j]);
          const float zp = fmaxf(z, 0.f) + 1e-5f;
          const float q = (z > 0.f) ? __builtin_amdgcn_rcpf(zp) : 0.f;     // torch.relu': 0 at z <= 0
          const float lz2 = __builtin_amdgcn_logf(zp);
          const float eu2 = (es2 - lb2[j]) + lz2;
          float aU;
          vc_obs<NOISE>(uv[j], eu2 * VC_LN2, eu2, rr[j], inv_s2_u, aU, ll[j], lt[j]);
          a += aU;
          w = aU * q;
          gau[j] += aU;
          gw[j] += w;
        }
        if (KIND != VC_KIND_VU) {
          const float wo = w * rec.omega;
          gnu[0][j] += a;
#pragma unroll
          for (int k = 0; k < H; ++k) {
            const float kk = (float)(k + 1);
            gnu[2 * k + 1][j] += a * rec.sn[k] + (FULL ? wo * kk * rec.cs[k] : 0.f);
            gnu[2 * k + 2][j] += a * rec.cs[k] - (FULL ? wo * kk * rec.sn[k] : 0.f);
          }
#pragma unroll
          for (int q = 0; q < NB; ++q) gnu[NH + q][j] += a * rec.db[q];
          A1 += a * dd;
        }
        if (FULL) A2 = fmaf(w, e2, A2);
        if (HAS_U) A3 += w * dd;
      }
      // per-cell sums over the 256 genes of this wave
      if (KIND == VC_KIND_PHASE) {
        const float t0 = vc_wave_sum(A1);
        keep0 = (lane == i) ? t0 : keep0;
      } else if (KIND == VC_KIND_VU) {
        const float t0 = vc_wave_sum(A3);
        keep0 = (lane == i) ? t0 : keep0;
      } else {
        const float t0 = vc_wave_sum(A1), t1 = vc_wave_sum(A2), t2 = vc_wave_sum(A3);
        keep0 = (lane == i) ? t0 : keep0;
        keep1 = (lane == i) ? t1 : keep1;
        keep2 = (lane == i) ? t2 : keep2;
      }
    }
    if (lane < n) {
      float* co = b.CO + ((size_t)gb * NCO) * d.Nc + cb + lane;
      co[0] = keep0;
      if (NCO == 3) { co[(size_t)d.Nc] = keep1; co[2 * (size_t)d.Nc] = keep2; }
    }
  }

  // ---- combine the 4 waves' gene-level partials through LDS, one store per workgroup ----------
  __shared__ float sm[VC_WAVES][NQ][VC_GBW];
  __shared__ float sm_ll[VC_WAVES];
  {
    float* row = &sm[wave][0][gl];
    auto put = [&](int q, const float* v) {
      *reinterpret_cast<float4*>(row + (size_t)q * VC_GBW) = make_float4(v[0], v[1], v[2], v[3]);
    };
    if (KIND == VC_KIND_VU) {
      put(0, gau); put(1, gw);
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) put(k, gnu[k]);
      // d loglik / d r (NB): -sum_c [log(r+mu) + (r+k)/(r+mu)] = -ln2 * sum log2 t - n_obs - (sum_c a)/r
      float gr[4];
      const float nobs = (float)(cend - cbeg) * (FULL ? 2.f : 1.f);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        gr[j] = (NOISE == VC_NOISE_NB) ? -VC_LN2 * lt[j] - nobs - gnu[0][j] * __builtin_amdgcn_rcpf(rr[j]) : 0.f;
      if (KIND == VC_KIND_PHASE) put(K, gr);
      else { put(K, gau); put(K + 1, gw); put(K + 2, gr); }
    }
    // likelihood partial in natural units: ln2 * (sum k (eta2 - log2 t) - r sum log2 t) for NB.
    // Padded genes are masked here (their nu~ is 0, so they never reached A1..A3).
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lj = VC_LN2 * ((NOISE == VC_NOISE_NB) ? ll[j] - rr[j] * lt[j] : ll[j]);
      l += (g0 + j < d.Ng) ? lj : 0.f;
    }
    l = vc_wave_sum(l);
    if (lane == 0) sm_ll[wave] = l;
  }
  __syncthreads();
  {
    const int t = threadIdx.x;   // gene t of the block
    float* go = b.GO + ((size_t)chunk * NQ) * d.Ng_pad + gb * VC_GBW + t;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      go[(size_t)q * d.Ng_pad] = (sm[0][q][t] + sm[1][q][t]) + (sm[2][q][t] + sm[3][q][t]);
    if (t == 0) b.LO[blockIdx.x] = (sm_ll[0] + sm_ll[1]) + (sm_ll[2] + sm_ll[3]);
  }
}

template <int H, int NB, int KIND, int NOISE>
static void vc_main_launch(const VcDims& d, const VcBufs& b, hipStream_t st) {
  hipLaunchKernelGGL((vc_main_kernel<H, NB, KIND, NOISE>), dim3(d.n_main_wg), dim3(256), 0, st, d, b);
}

struct VcMainEntry { int H, NB, kind, noise; vc_main_launch_fn fn; const void* kernel; };

// Explicit kernel instantiations are needed in both compilation passes; the launcher table is host-only.
#define VC_INST_K(KIND, NOISE, H, NB) \
  template __global__ void vc_main_kernel<H, NB, KIND, NOISE>(const VcDims, const VcBufs);
#define VC_INST_KROW(KIND, NOISE, H)                                                   \
  VC_INST_K(KIND, NOISE, H, 0) VC_INST_K(KIND, NOISE, H, 1) VC_INST_K(KIND, NOISE, H, 2) \
  VC_INST_K(KIND, NOISE, H, 3) VC_INST_K(KIND, NOISE, H, 4)
#define VC_ENT(KIND, NOISE, H, NB) \
  {H, NB, KIND, NOISE, &vc_main_launch<H, NB, KIND, NOISE>, (const void*)&vc_main_kernel<H, NB, KIND, NOISE>}
#define VC_ENT_ROW(KIND, NOISE, H)                                                          \
  VC_ENT(KIND, NOISE, H, 0), VC_ENT(KIND, NOISE, H, 1), VC_ENT(KIND, NOISE, H, 2), VC_ENT(KIND, NOISE, H, 3), \
  VC_ENT(KIND, NOISE, H, 4)

#if defined(__HIP_DEVICE_COMPILE__)
#define VC_DEFINE_TABLE(NAME, KIND, NOISE) \
  VC_INST_KROW(KIND, NOISE, 1) VC_INST_KROW(KIND, NOISE, 2) VC_INST_KROW(KIND, NOISE, 3)
#else
#define VC_DEFINE_TABLE(NAME, KIND, NOISE)                                                   \
  VC_INST_KROW(KIND, NOISE, 1) VC_INST_KROW(KIND, NOISE, 2) VC_INST_KROW(KIND, NOISE, 3)      \
  extern const VcMainEntry NAME[15] = {VC_ENT_ROW(KIND, NOISE, 1), VC_ENT_ROW(KIND, NOISE, 2), \
                                       VC_ENT_ROW(KIND, NOISE, 3)};
#endif
