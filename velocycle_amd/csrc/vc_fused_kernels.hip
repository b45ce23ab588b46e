// Fused single-rank SVI step (vc_svi_step_fused): three launches per step instead of four, and every O(Ng + Nc)
// latency chain of the step run once instead of twice.
//
//   K_main(t)    the likelihood kernel, unchanged (it also advances the device step counter: s = t + 1)
//   K_tail(t)    per gene block / per cell block, everything that is LOCAL to a gene or a cell, back to back in one
//                thread: second-stage reduction of K_main's partials -> chain rule to the parameter gradients (what
//                K_post does) -> pyro's ClippedAdam on exactly those parameters (what the optimiser kernel does) ->
//                the guide sample of step t + 1 from the fresh parameters, its prior / guide log-densities, the gene
//                table and the cell table of step t + 1 (what K_pre of the next step does)
//   K_omega(t)   the one GLOBAL dependency of the step: the angular-speed coefficients nu_omega.  Every block reduces
//                the per-block partials of d loglik / d nu_omega (a few hundred floats) redundantly, applies the
//                optimiser to those few parameters (block 0 stores them), samples nu_omega(t + 1) and fills
//                omega_c(t + 1) into the cell records of its 256 cells; block 0 also assembles the loss of step t
//                (fp64, fixed order); extra blocks evaluate the negative-binomial histogram terms for shape_inv(t + 1).
//
// The arithmetic of every piece is the one of vc_small_kernels.hip (K_pre / K_post / K_fin / ClippedAdam), statement by
// statement; tests/test_hip_fused.py holds the two paths against each other.  eps comes from the Philox stream only
// (seed, step, index): the host-eps parity path stays on the unfused kernels.  No Philox draw sits on the critical path:
// extra blocks of K_omega(t) draw the whole eps vector of step t + 2 into a three-slot ring (EPS[step % 3]) while the few
// blocks of the nu_omega chain run; K_tail / K_omega read the draws of the step being finished (for the scale gradients)
// and of the next sample from that ring -- slots that no block of the same launch writes.
// Reference semantics restated: velocity_inference_guide.py:9-141, phase_inference_guide.py:10-56, priors of
// velocity_inference_model.py:322-353,383 / phase_inference_model.py:360-366,392, pyro ClippedAdam.
// No compiler-chosen fused multiply-adds in this translation unit: the same source statement has to give the same bits in every
// kernel it is inlined into (K_post / K_fin, K_tail / K_omega, the sharded phases, the merged tail launch) -- hipcc's contraction
// of a * b + c depends on the surroundings of the statement.  Where a fused operation is wanted it is written as fmaf().
#pragma clang fp contract(off)
#include "vc_common.h"
#include "vc_tail_spec.h"

#define VC_PG_WAVES 16
#define VC_MAXQ (2 * VC_MAXH + 1 + VC_MAXNB + 3)
#define VC_MAXOWN 5      // parameters one (gene, role) thread owns (the LRMN cov_factor row is split over two roles)
#define VC_COVW 4        // cov_factor entries per role: role 14 holds k = 0..3, role 15 k = 4..7

struct VcOpt { float step_size, b1, b2, eps, clip, c2, wd; const unsigned char* frozen; };

#define VC_NWE (VC_MAX_NW * (VC_MAX_RANK + 2))      // nu_omega-related parameter elements at most
#define VC_HIST_ROUNDS 1     // one-launch tail, list form: rounds of 16 histogram tasks per 1024-thread block (2: measured slower)
#define VC_EPS_PER_THREAD 4  // one-launch tail: Philox pairs per thread of an eps block

// ---------------------------------------------------------------------------------------------
// Cells sharded over ranks (vc_svi_run_sharded): the same step cut at its ONE exchange.
//   K_main(t) -> K_tail phase A -> [sum of the exchange buffer over ranks] -> K_omega phase B        (3 launches + exchange)
// Phase A: gene blocks reduce K_main's partials and apply the chain rule, but write the gradient PARTIAL of this rank's cells
// into the exchange buffer X (same offsets as the gradient buffer) and stop; cell blocks are rank-local and run whole
// (phi_xy gradient, optimiser, next sample, cell record), their partials of d loglik / d nu_omega go to X's PW rows; one
// extra block folds the terms of this rank's loss that are complete before the launch (prior / guide terms of the sample,
// K_main's likelihood partials, the constant) into one double that crosses the exchange as four floats on fixed grids (vc_loss_split),
// the r-only likelihood terms of the gene blocks follow the same way.  Everything in X is additive over ranks; replicated prior terms carry root_w.
// Phase B (after the sum, one launch: gene blocks and K_omega's blocks do not depend on each other): gene blocks read the
// SUMMED gradient, apply ClippedAdam and draw the next sample; the nu_omega blocks reduce the summed PW rows (prior weight 1:
// the sum is complete); the loss block adds up the summed pairs; the histogram blocks re-derive the shape_inv update from a
// snapshot phase A took (the gene blocks of the same launch are rewriting it).
// ---------------------------------------------------------------------------------------------
#define VC_PH_ALL 0
#define VC_PH_A 1
#define VC_PH_B 2

// A rank's loss terms cross the exchange as floats and are ADDED there in float32 (RCCL ring / tree, or our own sums).  The
// terms cancel (likelihood partials and count constants of 1e7..1e9 against a total of 1e6), so a plain (hi, lo) split would
// lose ulps of the LARGE terms in the cross-rank add (observed: 1e-6 relative on the loss).  Each double therefore travels as
// four floats on FIXED grids -- multiples of 2^20, 2^1, 2^-17 and the remainder -- with fewer than 2^19 multiples each: the
// float32 sum of up to 16 ranks' pieces is exact piece by piece (|v| < 2^39), whatever the order of the adds.
#define VC_LOSS_PIECES 4
__device__ __forceinline__ void vc_loss_split(double v, float* __restrict__ out, int xmode = 0) {
  const double p0 = rint(v * (1.0 / 1048576.0)) * 1048576.0;
  const double r0 = v - p0;
  const double p1 = rint(r0 * 0.5) * 2.0;
  const double r1 = r0 - p1;
  const double p2 = rint(r1 * 131072.0) * (1.0 / 131072.0);
  vc_xstore(out, (float)p0, xmode); vc_xstore(out + 1, (float)p1, xmode); vc_xstore(out + 2, (float)p2, xmode);
  vc_xstore(out + 3, (float)(r1 - p2), xmode);
}
__device__ __forceinline__ double vc_loss_join(const float* __restrict__ in) {
  return (((double)in[0] + (double)in[1]) + (double)in[2]) + (double)in[3];
}
// the same on the exchange buffer (each piece summed over the ranks: by the exchange, or here -- vc_xget)
__device__ __forceinline__ double vc_loss_join_x(const VcXb& xb, long long i) {
  return (((double)vc_xget(xb, i) + (double)vc_xget(xb, i + 1)) + (double)vc_xget(xb, i + 2)) + (double)vc_xget(xb, i + 3);
}

// flat offset of element `ce` of angular-speed coefficient j: mean-field {loc, log scale}; LRMN tail row i = Ng + j
// {loc, R cov_factor entries, cov_diag}
__device__ __forceinline__ long long vc_nuw_elem_off(const VcDims& d, bool lrmn, int j, int ce) {
  if (!lrmn) return (ce == 0 ? d.poff[VC_P_NUOMEGA_LOCS] : d.poff[VC_P_NUOMEGA_USCALES]) + j;
  const long long i = (long long)d.Ng + j;
  return ce == 0 ? d.poff[VC_P_LRMN_LOC] + i
                 : (ce <= d.R ? d.poff[VC_P_LRMN_UCOV_FACTOR] + i * d.R + (ce - 1) : d.poff[VC_P_LRMN_UCOV_DIAG] + i);
}

// ---------------------------------------------------------------------------------------------
// K_tail, gene block: 1024 threads = 16 waves (roles) x 64 genes
// ---------------------------------------------------------------------------------------------
template <int MQ, int phase>
__device__ __forceinline__ void vc_tail_gene_block(const VcDims& d, const VcBufs& b, float* __restrict__ P,
                                                   float* __restrict__ G, float* __restrict__ Mm, float* __restrict__ Vv,
                                                   int header, int gblock, long long s, uint64_t seed, const VcOpt o,
                                                   int boot, const VcXb xb) {
  __shared__ float sm[VC_PG_WAVES][MQ][64];
  __shared__ double sm_ls[VC_PG_WAVES];
  __shared__ float sm_ws[2][64][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = gblock * 64 + lane;
  // waves 12..15 -> roles 12, 14, 15, 13: the heaviest role (13: log gamma / log beta or the LRMN core) sits on the SIMD
  // (wave % 4 == 3) whose other waves hold no nu[h] role, so that its serial chain does not share an issue port with them
  const int role = wave < 12 ? wave : (wave == 12 ? 12 : (wave == 15 ? 13 : wave + 1));
  const bool vel = d.model == VC_MODEL_VELOCITY;
  const bool lrmn = vel && d.guide == VC_GUIDE_LRMN;
  // the draws of the finished step (s - 1) and of the next sample (s): boot draws them itself (the ring is empty)
  const float* __restrict__ eps_new = b.EPS + (size_t)(s % 3) * d.eps_total;
  const float* __restrict__ eps_old = b.EPS + (size_t)((s + 2) % 3) * d.eps_total;
  auto draw_new = [&](long long idx) { return boot ? vc_philox_normal(seed, s, idx) : eps_new[idx]; };
  const bool nb = d.noise == VC_NOISE_NB;
  const int K = d.Kq, KT = d.K, Nh = d.Nh;      // K: coefficient rows of K_main's partials; KT: rows of the gene table in front of log beta
  const float rw = d.root_w;
  const size_t NP = d.Ng_pad;
  const bool live = g < d.Ng;
  // role kinds
  const bool r_nu = role < Nh;
  const bool r_dnu = !r_nu && role < Nh + d.Nb && d.with_dnu && !d.onehot;
  // one-hot batches: the waves of roles Nh .. 11 take the batch offsets q = role - Nh, + (12 - Nh), ... in a loop of their own
  // (vc_tail_dnu_onehot: any number of batches); the role machinery below sees no delta-nu role then
  const bool r_dnu1 = !r_nu && role < 12 && d.with_dnu && d.onehot;
  const bool r_si = role == 12 && nb;
  const bool r_mf = (role == 13 || role == 14) && vel && !lrmn;      // mean-field: role 13 log gamma, role 14 log beta
  const bool r_core = role == 13 && lrmn;
  const bool r_cov = (role == 14 || role == 15) && lrmn;
  const int kbase = (role - 14) * VC_COVW;            // first cov_factor column of a cov role
  const bool chain = !boot && phase != VC_PH_B;       // second-stage reduction + chain rule happen in this launch
  const bool upd = !boot && phase != VC_PH_A;         // ... the optimiser
  const bool samp = phase != VC_PH_A;                 // ... the next sample
  // (one-hot batches: the batch's chunk range is a scalar load the reduction's addresses depend on -- requested first, so that it
  // travels beside the roles' own inputs; two samples: the barrier was reached 1.1 us later without this)
  VcChunkWalk wk;
  wk.first = 0; wk.stride = 1; wk.end = 0;
  if (chain) wk = vc_chunk_walk(d, b, g, wave);

  VC_WSTAMP(0, 0);
  // ---- LRMN: eps_W of step s - 1 (gradient) and of step s (next sample) are wave-uniform: lane k fetches, readlane
  // broadcasts while every lane is active
  float ew_old[VC_MAX_RANK], ew_new[VC_MAX_RANK];
  {
    float mine_old = 0.f, mine_new = 0.f;
    if (r_cov && lane < d.R) {
      if (chain) mine_old = eps_old[d.eoff[VC_E_LRMN_W] + lane];
      if (samp) mine_new = draw_new(d.eoff[VC_E_LRMN_W] + lane);
    }
#pragma unroll
    for (int k = 0; k < VC_MAX_RANK; ++k) {
      ew_old[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine_old), k));
      ew_new[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine_new), k));
    }
  }

  // ---- owned parameters of this (gene, role) and everything else that does not depend on K_main's partials --------
  int off[VC_MAXOWN];
  float pp[VC_MAXOWN], pm[VC_MAXOWN], pv[VC_MAXOWN], gg[VC_MAXOWN];
  int nown = 0;
#pragma unroll
  for (int k = 0; k < VC_MAXOWN; ++k) { off[k] = 0; pp[k] = 0.f; pm[k] = 0.f; pv[k] = 0.f; gg[k] = 0.f; }
  constexpr int NIN = 12;
  float in[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) in[i] = 0.f;
  double HLg = 0.0, HDg = 0.0;
  float e0 = 0.f, e1 = 0.f;        // the Philox draws of step s this role needs
  long long jj = 0;                // flat index of the role's site element
  if (live) {
    if (r_nu) {
      jj = (long long)g * Nh + role;
      off[0] = (int)(d.poff[VC_P_NU_LOCS] + jj); off[1] = (int)(d.poff[VC_P_NU_USCALES] + jj); nown = 2;
      in[1] = b.sd_nu[jj]; in[2] = b.mu_nu[jj];
      if (chain && !CND(VC_SITE_NU)) { in[0] = b.lat[VC_SITE_NU][jj]; in[3] = eps_old[d.eoff[VC_E_NU] + jj]; }
      if (samp && !CND(VC_SITE_NU)) e0 = draw_new(d.eoff[VC_E_NU] + jj);   // a hidden site's draw is never used
    } else if (r_dnu) {
      jj = (long long)(role - Nh) * d.Ng + g;
      off[0] = (int)(d.poff[VC_P_DNU_LOCS] + jj); nown = 1;
      in[1] = vel ? 0.01f : b.sd_dnu[jj];
      if (chain && !CND(VC_SITE_DNU)) in[0] = b.lat[VC_SITE_DNU][jj];
    } else if (r_si) {
      off[0] = (int)(d.poff[VC_P_SHAPE_INV_ULOCS] + g); nown = 1;
      if (chain) {
        in[0] = b.GT[(size_t)(KT + 2) * NP + g];
        if (!CND(VC_SITE_SHAPE_INV)) in[1] = b.lat[VC_SITE_SHAPE_INV][g];
        {
          // the gene's histogram terms (usually 2..4 tasks): four requested per trip instead of one (a dependent round trip
          // each on this role's way to the barrier), the sums formed in task order as before.  They belong to the sample of
          // the step being finished: half (s - 1) & 1 when shape_inv is learned (the histogram blocks of THIS launch write
          // the other half)
          const double* __restrict__ HLs = b.HL + (size_t)(d.hist_par ? ((s - 1) & 1) : 0) * b.n_tasks;
          const double* __restrict__ HDs = b.HD + (size_t)(d.hist_par ? ((s - 1) & 1) : 0) * b.n_tasks;
          // (dense tables: one task per matrix and gene, no table of first tasks to wait for)
          const int nmat = vel ? 2 : 1;
          const int t0 = d.hist_dense ? nmat * g : b.h_tptr[g], t1 = d.hist_dense ? nmat * (g + 1) : b.h_tptr[g + 1];
          for (int tb = t0; tb < t1; tb += 4) {
            double hl[4], hd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int t = tb + k < t1 ? tb + k : tb; hl[k] = HLs[t]; hd[k] = HDs[t]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (tb + k < t1) { HLg += hl[k]; HDg += hd[k]; }
          }
        }
      }
    } else if (r_mf || r_core || r_cov) {
      in[2] = b.sd_g[g]; in[3] = b.mu_g[g]; in[5] = b.sd_b[g]; in[6] = b.mu_b[g];
      if (chain) {
        in[0] = b.GT[(size_t)(KT + 1) * NP + g];
        if (!CND(VC_SITE_LOGGAMMA)) in[1] = b.lat[VC_SITE_LOGGAMMA][g];
        if (!CND(VC_SITE_LOGBETA)) in[4] = b.lat[VC_SITE_LOGBETA][g];
      }
      if (r_mf) {
        if (role == 13) {
          off[0] = (int)(d.poff[VC_P_LOGGAMMA_LOCS] + g); off[1] = (int)(d.poff[VC_P_LOGGAMMA_USCALES] + g); nown = 2;
          if (chain) in[7] = eps_old[d.eoff[VC_E_LOGGAMMA] + g];
          if (samp) e0 = draw_new(d.eoff[VC_E_LOGGAMMA] + g);
        } else {
          off[0] = (int)(d.poff[VC_P_LOGBETA_LOCS] + g); off[1] = (int)(d.poff[VC_P_LOGBETA_USCALES] + g); nown = 2;
          if (chain) in[7] = eps_old[d.eoff[VC_E_LOGBETA] + g];
          if (samp) e0 = draw_new(d.eoff[VC_E_LOGBETA] + g);
        }
      } else {
        if (chain) { in[7] = b.lat_delta[g]; in[8] = b.lat_sgam[g]; }
        if (r_core) {
          off[0] = (int)(d.poff[VC_P_LOGBETA_LOCS] + g); off[1] = (int)(d.poff[VC_P_LOGBETA_USCALES] + g);
          off[2] = (int)(d.poff[VC_P_RHO_REAL_LOC] + g); off[3] = (int)(d.poff[VC_P_LRMN_LOC] + g);
          off[4] = (int)(d.poff[VC_P_LRMN_UCOV_DIAG] + g); nown = 5;
          if (chain) { in[9] = eps_old[d.eoff[VC_E_LOGBETA] + g]; in[10] = eps_old[d.eoff[VC_E_LRMN_D] + g]; }
          if (samp) { e0 = draw_new(d.eoff[VC_E_LRMN_D] + g); e1 = draw_new(d.eoff[VC_E_LOGBETA] + g); }
        } else {
#pragma unroll
          for (int k = 0; k < VC_COVW; ++k)
            if (kbase + k < d.R) off[k] = (int)(d.poff[VC_P_LRMN_UCOV_FACTOR] + (long long)g * d.R + kbase + k);
          nown = d.R - kbase < 0 ? 0 : (d.R - kbase > VC_COVW ? VC_COVW : d.R - kbase);
          if (chain) { in[9] = P[d.poff[VC_P_LOGBETA_USCALES] + g]; in[10] = P[d.poff[VC_P_RHO_REAL_LOC] + g]; }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < VC_MAXOWN; ++k)
      if (k < nown) {
        pp[k] = P[off[k]];
        if (upd) { pm[k] = Mm[off[k] - header]; pv[k] = Vv[off[k] - header]; }
        if (upd && phase == VC_PH_B) gg[k] = vc_xget(xb, off[k]);   // the gradient summed over ranks
      }
    if (r_si && !upd && samp) { pm[0] = Mm[off[0] - header]; pv[0] = Vv[off[0] - header]; }     // boot: for the snapshot below
    // phase A: the histogram blocks of phase B re-derive the shape_inv update while the gene blocks of the same launch
    // rewrite it -- they read this snapshot {parameter, exp_avg, exp_avg_sq}
    if (phase == VC_PH_A && r_si) {
      vc_xstore(xb.sis + g, pp[0], xb.xmode);
      vc_xstore(xb.sis + NP + g, Mm[off[0] - header], xb.xmode);
      vc_xstore(xb.sis + 2 * NP + g, Vv[off[0] - header], xb.xmode);
    }
  }

  // one-hot batches: the chunk range of this wave's first batch (a dependent fetch in front of its rows: asked for with the rest)
  VcDnuPre dnu_pre;
  const bool dnu_lds = vc_walk_by_batch(d);      // the batches' sums come out of the second-stage reduction itself (vc_chunk_walk)
  const bool dnu_pre_on = chain && r_dnu1 && live && role - Nh < d.Nb && !CND(VC_SITE_DNU) && !dnu_lds;
  dnu_pre.c0 = 0; dnu_pre.c1 = 0;
  float dq_p = 0.f, dq_m = 0.f, dq_v = 0.f, dq_lat = 0.f;      // ... and that batch's parameter, moments and sample (as the roles above)
  // conditioned batch offsets (the tutorials' velocity stage hands over the phase fit's delta nu: Tutorial_Aissa_PC9_TwoSample cell 42)
  // never change: parameter, moments, site value and gene-table row hold what the priming launch wrote -- only their (constant)
  // prior term is re-formed per step; no load / optimiser / store round trip on these waves (round 6: the two-sample tail gap)
  const bool dnu_fixed = CND(VC_SITE_DNU) && !boot && phase != VC_PH_A;
  const bool dq_on = r_dnu1 && live && role - Nh < d.Nb && !dnu_fixed;
  if (dq_on) {
    const long long jq0 = (long long)(role - Nh) * d.Ng + g;
    const int po0 = (int)(d.poff[VC_P_DNU_LOCS] + jq0);
    dq_p = P[po0];
    if (upd) { dq_m = Mm[po0 - header]; dq_v = Vv[po0 - header]; }
    if (chain && !CND(VC_SITE_DNU)) dq_lat = b.lat[VC_SITE_DNU][jq0];
  }
  if (dnu_pre_on) {
    const int gbm = g / d.gbw;
    dnu_pre.c0 = b.bat_chunk[gbm * (d.Nb + 1) + role - Nh];
    dnu_pre.c1 = b.bat_chunk[gbm * (d.Nb + 1) + role - Nh + 1];
  }
  VC_WSTAMP(0, 1);
  double loss_post = 0.0;
  if (chain) {
    // ---- second-stage reduction of K_main's gene-level partials (as K_post) ------------------------------------
    constexpr int U = MQ <= 2 ? 16 : (MQ <= 4 ? 12 : (MQ <= 6 ? 8 : 2));     // chunk groups in flight per wave (register budget: 128)
    float acc[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) acc[q] = 0.f;
    for (int ch0 = wk.first; ch0 < wk.end; ch0 += U * wk.stride) {
      float v[U][MQ];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ch = ch0 + u * wk.stride;
        const float* go = b.GO + ((size_t)(ch < wk.end ? ch : ch0) * d.nq) * NP + g;
#pragma unroll
        for (int q = 0; q < MQ; ++q) v[u][q] = (q < d.nq) ? go[(size_t)q * NP] : 0.f;
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
    // one-hot batches: the rows of the first batch of this wave's loop further down are REQUESTED here (the batch's chunk range was
    // fetched at the top), nothing is consumed: the barrier is not held up, and the rows are there when the loop wants them
    // (summing them in front of the barrier held every role of the block back by 3 us: profiles/r05_tail_spec.md)
    if (dnu_pre_on) vc_dnu_range_issue(d, b, g, dnu_pre);
    // Everything requested at the top of the block must have ARRIVED before this barrier: several roles read values another role of
    // the same gene rewrites behind it (the LRMN cov roles read log-scale and rho of the core role's parameters, every velocity role
    // reads gamma and the samples of log gamma / log beta) -- correct only if the read happens in front of the barrier.  `P` and the
    // tables are __restrict__ / plain pointers the compiler may read as late as the first use, i.e. BEHIND the barrier, next to the
    // other wave's store (round 5: vel_lrmn_joint lost its 2e-10 agreement with the unfused sequence in one build of this file and
    // kept it in the next).  An empty asm that takes the values as operands pins the loads here.
    // (only the roles that read what another role owns: 13..15; their own parameters and moments need no pin)
    if (role >= 13) asm volatile("" ::"v"(in[0]), "v"(in[1]), "v"(in[4]), "v"(in[7]), "v"(in[8]), "v"(in[9]), "v"(in[10]));
    VC_WSTAMP(0, 2);
    __syncthreads();
    VC_WSTAMP(0, 3);
    auto T = [&](int q) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < VC_PG_WAVES; ++w) t += sm[w][q][lane];
      return t;
    };
    // ---- chain rule to the gradients of the owned parameters (statement by statement vc_post_gene_block) -------
    if (live) {
      if (r_nu) {
        if (!CND(VC_SITE_NU)) {
          const float x = in[0], sd = in[1];
          const float gx = T(role) - rw * (x - in[2]) / (sd * sd);
          gg[0] = -gx;
          gg[1] = -gx * expf(pp[1]) * in[3] - rw;
        }
      } else if (r_dnu) {
        if (!CND(VC_SITE_DNU)) {
          const float x = in[0], sd = in[1];
          gg[0] = -(T(role) - rw * x / (sd * sd));
        }
      } else if (r_si) {
        const float r = in[0];
        const float U_r = (d.kind == VC_KIND_PHASE) ? T(K) : (d.kind == VC_KIND_VFULL ? T(K + 2) : 0.f);
        // (single rank: the r-only loss terms are split -- LPR was written when this sample was drawn, the histogram sums are
        // added by the loss block; the sharded step sends them over the exchange per gene block)
        if (phase == VC_PH_A && d.nmat_r > 0) loss_post -= (double)d.nmat_r * d.Nc * (double)r * (double)logf(r) + HLg;
        if (!CND(VC_SITE_SHAPE_INV)) gg[0] = vc_si_grad(d, r, in[1], U_r, HDg, rw);
      } else if (r_mf || r_core || r_cov) {
        const float gam = in[0];
        float U_lb, U_lg;
        if (d.kind == VC_KIND_VFULL) { U_lb = -T(K); U_lg = T(K + 1) * gam; }
        else { U_lb = -T(0); U_lg = T(1) * gam; }
        float g_lg = 0.f, g_lb = 0.f;
        if (!CND(VC_SITE_LOGGAMMA)) g_lg = U_lg - rw * (in[1] - in[3]) / (in[2] * in[2]);
        if (!CND(VC_SITE_LOGBETA)) g_lb = U_lb - rw * (in[4] - in[6]) / (in[5] * in[5]);
        if (r_mf) {
          if (role == 13) {
            const bool cg = CND(VC_SITE_LOGGAMMA);
            gg[0] = -g_lg;
            gg[1] = cg ? 0.f : -g_lg * expf(pp[1]) * in[7] - rw;
          } else {
            const bool cb = CND(VC_SITE_LOGBETA);
            gg[0] = -g_lb;
            gg[1] = cb ? 0.f : -g_lb * expf(pp[1]) * in[7] - rw;
          }
        } else {
          const bool cb = CND(VC_SITE_LOGBETA);
          const float A = g_lb;
          const float ent = cb ? 0.f : rw;
          const float delta = in[7], sgam = in[8];
          const float sb = expf(r_core ? pp[1] : in[9]);
          const float rho_real = r_core ? pp[2] : in[10];
          const float sg = sigmoidf_(rho_real / d.rho_scale);
          const float rho = sg * 1.998f - 0.999f;
          const float om = 1.f - rho * rho, sq = sqrtf(om);
          const float dl_ddelta = -g_lg - A * rho * sb / sgam;
          const float dl_dsg = A * rho * sb * delta / (sgam * sgam);
          if (r_core) {
            const float eb = in[9];
            gg[0] = -A;
            gg[1] = -A * (rho * delta / sgam + sq * eb) * sb - ent;
            float g_rho = -A * (sb * delta / sgam - sb * rho * eb / sq) + ent * rho / om;
            float g_rr = g_rho * 1.998f * sg * (1.f - sg) / d.rho_scale;
            if (!CND(VC_SITE_RHO_REAL)) g_rr += rw * (rho_real - d.rho_mean) / (d.rho_std * d.rho_std);
            gg[2] = g_rr;
            gg[3] = -g_lg;
            const float dg = expf(pp[4]);
            const float ed = in[10];
            gg[4] = (dl_ddelta * ed / (2.f * sqrtf(dg)) + dl_dsg / (2.f * sgam)) * dg;
          } else {
#pragma unroll
            for (int k = 0; k < VC_COVW; ++k)
              if (k < nown) {
                const float w = expf(pp[k]);
                const float ew = kbase ? ew_old[VC_COVW + k] : ew_old[k];
                gg[k] = (w > 0.f) ? (dl_ddelta * ew + dl_dsg * w / sgam) * w : 0.f;
              }
          }
        }
      }
      // ---- phase A: this rank's gradient partial into the exchange buffer ----------------------------------------
      if (phase == VC_PH_A) {
#pragma unroll
        for (int k = 0; k < VC_MAXOWN; ++k)
          if (k < nown) vc_xput(xb, off[k], gg[k]);
      }
    }
    if (phase == VC_PH_A && role == 12) {
      const double tot = vc_wave_sum_d63(loss_post);
      if (lane == 63) {
        b.LPP[gblock] = tot;
        vc_loss_split(tot, xb.x + xb.loss_off + VC_LOSS_PIECES * (1 + gblock), xb.xmode);
      }
    }
  }
  // ---- one-hot batches: delta nu[q, g] of this wave's batches, start to end (gradient from the batch's workgroups -> phase A's
  // partial / ClippedAdam -> the value of the next sample, its prior term, gene-table row) -------------------------------
  float logp_dnu = 0.f;
  if (r_dnu1 && live) {
    for (int q = role - Nh; q < d.Nb; q += 12 - Nh) {
      const long long jq = (long long)q * d.Ng + g;
      const int po = (int)(d.poff[VC_P_DNU_LOCS] + jq);
      const float sd = vel ? 0.01f : b.sd_dnu[jq];
      const bool first = q == role - Nh;          // (its inputs were requested at the top of the block)
      if (dnu_fixed) {
        if (samp) logp_dnu += vc_normal_lp(b.cnd[VC_SITE_DNU][jq], 0.f, sd);
        continue;
      }
      float p = first ? dq_p : P[po];
      if (!boot) {
        float gq = 0.f;
        if (phase == VC_PH_B) gq = vc_xget(xb, po);                    // the gradient summed over ranks
        else if (!CND(VC_SITE_DNU)) {
          float lik;
          if (dnu_lds) {          // the partials of this batch's waves, in wave order (row 0 = the constant harmonic)
            int w0, nw;
            vc_walk_waves_of(d, q, &w0, &nw);
            lik = 0.f;
            for (int w = w0; w < w0 + nw; ++w) lik += sm[w][0][lane];
          } else {
            lik = (dnu_pre_on && first) ? vc_dnu_range_finish(d, b, g, dnu_pre) : vc_dnu_range_sum(d, b, g, q);
          }
          gq = -(lik - rw * (first ? dq_lat : b.lat[VC_SITE_DNU][jq]) / (sd * sd));
        }
        if (phase == VC_PH_A) vc_xput(xb, po, gq);
        else {
          G[po] = gq;
          float mm = first ? dq_m : Mm[po - header], vv = first ? dq_v : Vv[po - header];
          p = vc_adam_elem(p, gq, mm, vv, o.step_size, o.b1, o.b2, o.eps, o.clip, o.c2, vc_wd_at(o.wd, o.frozen, po));
          Mm[po - header] = mm; Vv[po - header] = vv; P[po] = p;
        }
      }
      if (samp) {
        const float x = CND(VC_SITE_DNU) ? b.cnd[VC_SITE_DNU][jq] : p;
        logp_dnu += vc_normal_lp(x, 0.f, sd);
        b.lat[VC_SITE_DNU][jq] = x;
        b.GT[(size_t)(Nh + q) * NP + g] = x;
      }
    }
  }
  // ---- gradient out, ClippedAdam on the owned parameters (phase B: gg is the sum over ranks) -------------------------
  if (upd && live) {
#pragma unroll
    for (int k = 0; k < VC_MAXOWN; ++k)
      if (k < nown) {
        G[off[k]] = gg[k];
        const float np = vc_adam_elem(pp[k], gg[k], pm[k], pv[k], o.step_size, o.b1, o.b2, o.eps, o.clip, o.c2, vc_wd_at(o.wd, o.frozen, off[k]));
        Mm[off[k] - header] = pm[k];
        Vv[off[k] - header] = pv[k];
        P[off[k]] = np;
        pp[k] = np;
      }
  }
  if (!samp) return;                    // phase A ends here (uniform per launch: no barrier is skipped by part of a block)

  VC_WSTAMP(0, 4);
  // ---- the guide sample of step s from the fresh parameters (statement by statement vc_pre_kernel) ---------------
  float logp = logp_dnu, logq = 0.f;
  double lpr = 0.0;
  if (g < d.Ng_pad && !live) {
    if (boot && wave == 0) {       // padded gene: nu~ = 0 (never reaches a per-cell sum), loss masked in K_main
      float* GT = b.GT + g;
      for (int k = 0; k < KT; ++k) GT[k * NP] = 0.f;
      GT[KT * NP] = 0.f; GT[(KT + 1) * NP] = 1.f; GT[(KT + 2) * NP] = 1.f;
    }
  }
  if (live) {
    float* GT = b.GT + g;
    if (r_nu) {
      const float e = e0;
      float x;
      if (CND(VC_SITE_NU)) x = b.cnd[VC_SITE_NU][jj];
      else {
        const float u = pp[1];
        x = pp[0] + expf(u) * e;
        logq += -0.5f * e * e - u - 0.5f * VC_LOG_2PI;
      }
      logp += vc_normal_lp(x, in[2], in[1]);
      b.lat[VC_SITE_NU][jj] = x;
      GT[role * NP] = x;
    } else if (r_dnu) {
      const float x = CND(VC_SITE_DNU) ? b.cnd[VC_SITE_DNU][jj] : pp[0];
      logp += vc_normal_lp(x, 0.f, in[1]);
      b.lat[VC_SITE_DNU][jj] = x;
      GT[role * NP] = x;
    } else if (r_si) {
      const float si = CND(VC_SITE_SHAPE_INV) ? b.cnd[VC_SITE_SHAPE_INV][g] : expf(pp[0]);
      logp += d.gamma_alpha * logf(d.gamma_beta) + (d.gamma_alpha - 1.f) * logf(si) - d.gamma_beta * si - d.lgamma_alpha;
      b.lat[VC_SITE_SHAPE_INV][g] = si;
      const float r = 1.0f / si;
      GT[(KT + 2) * NP] = r;
      // the r-only likelihood term of THIS sample, for the loss of the step it belongs to (half s & 1)
      if (d.nmat_r > 0) lpr = -((double)d.nmat_r * d.Nc * (double)r * (double)logf(r));
      if (!CND(VC_SITE_SHAPE_INV)) {      // what the histogram blocks of the next launch re-derive the update from
        float* sis = b.SIS + (size_t)(s & 1) * 4 * NP + g;
        sis[0] = pp[0]; sis[NP] = pm[0]; sis[2 * NP] = pv[0]; sis[3 * NP] = si;
      }
    } else if (role == 12 && boot && !nb) {
      GT[(KT + 2) * NP] = 1.0f;
    } else if (r_mf && role == 13) {
      const float eg = e0;
      const float ug = pp[1];
      const float lg_guide = pp[0] + expf(ug) * eg;
      if (!CND(VC_SITE_LOGGAMMA)) logq += -0.5f * eg * eg - ug - 0.5f * VC_LOG_2PI;
      const float lg = CND(VC_SITE_LOGGAMMA) ? b.cnd[VC_SITE_LOGGAMMA][g] : lg_guide;
      logp += vc_normal_lp(lg, in[3], in[2]);
      b.lat[VC_SITE_LOGGAMMA][g] = lg;
      GT[(KT + 1) * NP] = expf(lg);
    } else if (r_mf) {
      const float eb = e0;
      const float ub = pp[1];
      const float lb_guide = pp[0] + expf(ub) * eb;
      if (!CND(VC_SITE_LOGBETA)) logq += -0.5f * eb * eb - ub - 0.5f * VC_LOG_2PI;
      const float lbv = CND(VC_SITE_LOGBETA) ? b.cnd[VC_SITE_LOGBETA][g] : lb_guide;
      logp += vc_normal_lp(lbv, in[6], in[5]);
      b.lat[VC_SITE_LOGBETA][g] = lbv;
      GT[KT * NP] = lbv;
    } else if (r_cov) {
      // LowRankMultivariateNormal.rsample, low-rank part: sum_k W[g,k] eps_W[k] and sum_k W[g,k]^2 -> role 13
      float dW = 0.f, w2 = 0.f;
#pragma unroll
      for (int k = 0; k < VC_COVW; ++k)
        if (k < nown) {
          const float w = expf(pp[k]);
          dW += w * (kbase ? ew_new[VC_COVW + k] : ew_new[k]);
          w2 += w * w;
        }
      sm_ws[role - 14][lane][0] = dW;
      sm_ws[role - 14][lane][1] = w2;
    }
  }
  if (lrmn) __syncthreads();
  if (live && r_core) {
    float* GT = b.GT + g;
    const float ed = e0, eb = e1;
    // columns 0..3 summed by role 14, 4..7 by role 15: for rank <= 5 (the reference's default) the same order of
    // additions as the sequential loop of vc_pre_kernel
    float delta = sm_ws[0][lane][0];
    float w2 = sm_ws[0][lane][1];
    if (d.R > VC_COVW) { delta += sm_ws[1][lane][0]; w2 += sm_ws[1][lane][1]; }
    const float dg = expf(pp[4]);
    delta += sqrtf(dg) * ed;
    const float sgam = sqrtf(w2 + dg);
    const float lg_guide = pp[3] + delta;
    const float rho_real_g = pp[2];
    const float rho = sigmoidf_(rho_real_g / d.rho_scale) * 1.998f - 0.999f;
    const float ub = pp[1];
    const float sb = expf(ub);
    const float tt = sb * sqrtf(1.f - rho * rho);
    const float lb_guide = pp[0] + rho * sb * delta / sgam + tt * eb;
    if (!CND(VC_SITE_LOGBETA)) logq += -0.5f * eb * eb - logf(tt) - 0.5f * VC_LOG_2PI;
    b.lat_delta[g] = delta;
    b.lat_sgam[g] = sgam;
    const float rho_val = CND(VC_SITE_RHO_REAL) ? b.cnd[VC_SITE_RHO_REAL][g] : rho_real_g;
    logp += vc_normal_lp(rho_val, d.rho_mean, d.rho_std);
    b.lat[VC_SITE_RHO_REAL][g] = rho_val;
    const float lg = CND(VC_SITE_LOGGAMMA) ? b.cnd[VC_SITE_LOGGAMMA][g] : lg_guide;
    const float lbv = CND(VC_SITE_LOGBETA) ? b.cnd[VC_SITE_LOGBETA][g] : lb_guide;
    logp += vc_normal_lp(lg, in[3], in[2]) + vc_normal_lp(lbv, in[6], in[5]);
    b.lat[VC_SITE_LOGGAMMA][g] = lg;
    b.lat[VC_SITE_LOGBETA][g] = lbv;
    GT[KT * NP] = lbv;
    GT[(KT + 1) * NP] = expf(lg);
  }
  if (role == 12) {                                // (every gene block has this wave; 0 without a negative binomial)
    const double tot = vc_wave_sum_d63(lpr);
    if (lane == 63) b.LPR[(size_t)(s & 1) * d.nb_post_gene + gblock] = tot;
  }
  if (live && !vel && role == 13 && boot) {      // phase model: the velocity rows of the gene table are constants
    b.GT[(size_t)KT * NP + g] = 0.f;
    b.GT[(size_t)(KT + 1) * NP + g] = 1.f;
  }
  VC_WSTAMP(0, 5);
  // prior / guide terms of the step-s sample: fp64 block sum in fixed order
  {
    const double lt = live ? -(double)rw * ((double)logp - (double)logq) : 0.0;
    const double ws = vc_wave_sum_d63(lt);
    if (lane == 63) sm_ls[wave] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < VC_PG_WAVES; ++w) t += sm_ls[w];
      b.LPF[(size_t)(s & 1) * d.nlpf + gblock] = t;
    }
  }
  VC_WSTAMP(0, 6);
}

// ---------------------------------------------------------------------------------------------
// K_tail, cell block: d.tail_tc cells.  256 cells on the first 4 waves of the (1024-thread) block -- one wave per SIMD: the
// per-cell chain (Philox, Adam, atan2, sincos) is instruction-bound, 16 waves on one CU would serialise it four deep; large
// shards (vc_engine.hip) use all 16 waves, 1024 cells per block, because the number of blocks to place then dominates
// ---------------------------------------------------------------------------------------------
#define VC_TC_MAX 1024
struct VcNuwShared {     // LDS of the nu_omega chain (vc_nuw_chain, below)
  float up[VC_MAX_NW];
  float np[VC_NWE];
  float nuw[VC_MAX_NW];
  double lq[VC_MAX_NW];
};
#define VC_NUW_RAW 12          // rows per lane held in registers: n_pw <= 768 partial rows (3 workgroups per CU x 256 CUs)
struct VcNuwRaw { float r[2][VC_NUW_RAW]; };
__device__ __forceinline__ void vc_nuw_sums_issue(const VcDims& d, const VcBufs& b, int boot, int phase, const VcXb& xb, int nthr,
                                                  VcNuwRaw& raw);
__device__ __forceinline__ void vc_nuw_sums_finish(const VcDims& d, const VcBufs& b, int boot, int phase, const VcXb& xb, int nthr,
                                                   const VcNuwRaw& raw, VcNuwShared& sh);
__device__ __forceinline__ void vc_nuw_chain(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                             long long s, uint64_t seed, const VcAdamArgs& a, int boot, bool first, int phase,
                                             const VcXb& xb, int c, float s1, float c1, int nthr, VcNuwShared& sh);
__device__ __forceinline__ void vc_nuw_wave(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                            long long s, uint64_t seed, const VcAdamArgs& a, bool first, VcNuwShared& sh);
__device__ __forceinline__ void vc_nuw_cells(const VcDims& d, const VcBufs& b, int c, float s1, float c1, const VcNuwShared& sh);
// OMEGA: the nu_omega chain of K_omega runs INSIDE this block (every cell block redundantly, block 0 stores), on K_main's own
// partials (pw_inline) -- the cell record of step s leaves this block complete, omega_c included (vc_tail2_kernel)
template <int phase, bool OMEGA = false>
__device__ __forceinline__ void vc_tail_cell_block(const VcDims& d, const VcBufs& b, float* __restrict__ P,
                                                   float* __restrict__ G, float* __restrict__ Mm, float* __restrict__ Vv,
                                                   int header, int cblock, long long s, uint64_t seed, const VcOpt o,
                                                   int boot, const VcXb xb) {
  __shared__ float sm_w[VC_TC_MAX / 64][VC_MAX_NW];
  __shared__ double sm_lc[VC_TC_MAX / 64];
  const int VC_TC = d.tail_tc;
  const bool vel = d.model == VC_MODEL_VELOCITY;
  // OMEGA, blocks with a wave to spare (<= 960 cells): the nu_omega chain -- sums of K_main's partials, gradient, optimiser, next
  // sample -- depends on nothing the cell waves do, so ONE extra wave (the "chain wave", right behind the cell waves) runs it
  // BESIDE them; the cell waves meet it at one barrier and only add omega_c to their records (round 5: the chain used to start
  // when the cell part was done, 4 us of barriers and dependent exp / sqrt / divide at the end of the launch's longest block)
#ifdef VC_NO_CHAIN_WAVE
  const bool cwm = false;
#else
  const bool cwm = OMEGA && vel && !boot && d.pw_inline && VC_TC <= VC_TC_MAX - 64;
#endif
  if ((int)threadIdx.x >= VC_TC + (cwm ? 64 : 0)) return;   // 256-cell blocks: the other waves of the block have nothing to do
  VC_WSTAMP(0, 0);
  __shared__ VcNuwShared sh_nuw;
  if (cwm && (int)threadIdx.x >= VC_TC) {
    vc_nuw_wave(d, b, P, G, s, seed, VcAdamArgs{Mm, Vv, 0.0, 0.0, 0.0, 0.0, o.b1, o.b2, o.eps, o.clip, header, o.wd, 0, o.frozen}, cblock == 0, sh_nuw);
    __syncthreads();          // sh_nuw.nuw is complete: the cell waves take it from here
    __syncthreads();          // (the block's last barrier, below)
    return;
  }
  // OMEGA without a chain wave: the sums of K_main's partials of d loglik / d nu_omega depend on nothing in this block -- requested
  // first, so that their round trip runs beside the cell part's
  VcNuwRaw nuw_raw;
  if (OMEGA && vel && !cwm) vc_nuw_sums_issue(d, b, boot, phase, xb, VC_TC, nuw_raw);
  const int c = cblock * VC_TC + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool in_range = c < d.Nc;
  const bool cxy = CND(VC_SITE_PHIXY);
  float A[3] = {0.f, 0.f, 0.f};
  float dx01[2] = {0.f, 0.f};
  float2 pp = make_float2(0.f, 0.f), pm = pp, pv = pp, pxy = pp;
  float2 sc_old = make_float2(0.f, 1.f);          // sin, cos of the phase of the step being finished (cell record)
  float ex = 0.f, ey = 0.f;
  double loss = 0.0;
  const long long poff = d.poff[VC_P_PHIXY_LOCS] + 2LL * c;
  const int cp = in_range ? vc_pos(b, c) : 0;      // the cell's place in the likelihood kernel's order (record, partial rows, W row)
  const float2* ctr = reinterpret_cast<const float2*>(b.CT + (size_t)cp * d.ctw);
  if (in_range) {
    pxy = *reinterpret_cast<const float2*>(b.pxy + 2 * (size_t)c);
    if (!cxy) {
      pp = *reinterpret_cast<const float2*>(P + poff);
      // the ring holds this rank's slice of the stream (local index); boot draws directly at the GLOBAL index ((x, y) of a
      // cell are the two normals of one Philox block: the eps layout starts phi_xy at an even index)
      const long long gi = d.eoff[VC_E_PHIXY] + 2LL * c;
      if (boot) vc_philox_normal2(seed, s, (uint64_t)(gi + 2LL * d.cell_offset) >> 1, ex, ey);
      else {
        const float2 e2 = *reinterpret_cast<const float2*>(b.EPS + (size_t)(s % 3) * d.eps_total + gi);
        ex = e2.x; ey = e2.y;
      }
    }
    if (!boot) {
      float2 xy = make_float2(1.f, 0.f);
      float om = 0.f, dom = 0.f;
      if (vel) {
        sc_old = make_float2(ctr[0].x, ctr[1].x);
        dx01[0] = b.Dm[c];
        if (d.Nx > 1) dx01[1] = b.Dm[(size_t)d.Nc + c];
      }
      if (!cxy) {
        pm = *reinterpret_cast<const float2*>(Mm + (poff - header));
        pv = *reinterpret_cast<const float2*>(Vv + (poff - header));
        xy = *reinterpret_cast<const float2*>(b.lat[VC_SITE_PHIXY] + 2 * (size_t)c);
        if (d.kind == VC_KIND_VFULL) { om = b.lat_omega[c]; dom = b.lat_domega[c]; }
      }
      // K_main's per-cell partial rows, four gene blocks requested per trip before the first is added (an `A += load` loop with
      // run-time bounds is one dependent round trip per row: 12 of them for the S+U kernel's three rows -- 6 us of this block's
      // chain, profiles/r04_two_launch.md); every A[j] still adds its rows in gene-block order
      for (int gb0 = 0; gb0 < d.nGB; gb0 += 4) {
        float v[4][3];
        if (d.nco == 3) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 3; ++j) v[u][j] = b.CO[((size_t)(gb0 + u < d.nGB ? gb0 + u : gb0) * 3 + j) * d.Nc + cp];
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            v[u][0] = b.CO[(size_t)(gb0 + u < d.nGB ? gb0 + u : gb0) * d.Nc + cp];
            v[u][1] = 0.f; v[u][2] = 0.f;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (gb0 + u < d.nGB) {
#pragma unroll
            for (int j = 0; j < 3; ++j) if (j < d.nco) A[j] += v[u][j];
          }
      }
      if (!cxy) {
        float dphi = A[0];
        if (d.kind == VC_KIND_VFULL) dphi += om * A[1] + A[2] * dom;
        const float x = xy.x, y = xy.y;
        const float inv = 1.0f / (x * x + y * y);
        const float gx = -(dphi * (-y * inv) - (x - pxy.x));
        const float gy = -(dphi * (x * inv) - (y - pxy.y));
        *reinterpret_cast<float2*>(G + poff) = make_float2(gx, gy);
        pp.x = vc_adam_elem(pp.x, gx, pm.x, pv.x, o.step_size, o.b1, o.b2, o.eps, o.clip, o.c2, vc_wd_at(o.wd, o.frozen, poff));
        pp.y = vc_adam_elem(pp.y, gy, pm.y, pv.y, o.step_size, o.b1, o.b2, o.eps, o.clip, o.c2, vc_wd_at(o.wd, o.frozen, poff));
        *reinterpret_cast<float2*>(Mm + (poff - header)) = pm;
        *reinterpret_cast<float2*>(Vv + (poff - header)) = pv;
        *reinterpret_cast<float2*>(P + poff) = pp;
      } else {
        // conditioned phases: the gradient is zero, so are the moments -- ClippedAdam leaves the parameter where it is
        *reinterpret_cast<float2*>(G + poff) = make_float2(0.f, 0.f);
      }
    }
  }
  VC_WSTAMP(0, 3);
  if (OMEGA && vel && !cwm) vc_nuw_sums_finish(d, b, boot, phase, xb, VC_TC, nuw_raw, sh_nuw);      // (their rows arrived with the cell's own loads)
  // (single rank with K_main's own partials: nobody reads the cell blocks' -- K_omega / the chain below take PWM)
  if (vel && !boot && !(phase == VC_PH_ALL && d.pw_inline)) {
    // partial sums of d loglik / d nu_omega[x,h] = sum_c A3_c D[x,c] zeta_omega_h(phi_c) at the phases of step s - 1
    // (their sin / cos are in the cell record; higher harmonics by the angle-addition recurrence, as K_pre built them)
    const float a3 = in_range ? (d.kind == VC_KIND_VFULL ? A[2] : A[0]) : 0.f;
    const float s1 = sc_old.x, c1 = sc_old.y;
    float sk[VC_MAXH], ck[VC_MAXH];
    sk[0] = s1; ck[0] = c1;
    for (int k = 1; k < d.Hw && k < VC_MAXH; ++k) {
      sk[k] = sk[k - 1] * c1 + ck[k - 1] * s1;
      ck[k] = ck[k - 1] * c1 - sk[k - 1] * s1;
    }
    for (int xq = 0; xq < d.Nx; ++xq) {
      const float dx = in_range ? (xq < 2 ? dx01[xq] : b.Dm[(size_t)xq * d.Nc + c]) : 0.f;
      for (int h = 0; h < d.Nhw; ++h) {
        const float z = (h == 0) ? 1.f : ((h & 1) ? sk[(h - 1) >> 1] : ck[(h - 1) >> 1]);
        const float t = vc_wave_sum(a3 * dx * z);
        if (lane == 0) sm_w[wave][xq * d.Nhw + h] = t;
      }
    }
  }
  // ---- snapshot of the nu_omega-related parameters, their moments and the nu_omega value of the step being finished:
  // K_omega's blocks all READ this copy while its block 0 stores the updated values (no reader ever races the writer)
  if (vel && cblock == 0 && !OMEGA) {      // (OMEGA: the chain's block 0 of the launch before wrote this copy itself)
    const bool lrmn = d.guide == VC_GUIDE_LRMN;
    const int fin_per = lrmn ? d.R + 2 : 2;
    float* __restrict__ nws = b.NWS + (size_t)(s & 1) * 4 * VC_NWE;      // the copy K_omega reads at this step (two: by parity)
    for (int tt = threadIdx.x; tt < d.NW * fin_per; tt += VC_TC) {
      const long long off = vc_nuw_elem_off(d, lrmn, tt / fin_per, tt % fin_per);
      nws[tt] = P[off];
      nws[VC_NWE + tt] = Mm[off - header];
      nws[2 * VC_NWE + tt] = Vv[off - header];
    }
    if ((int)threadIdx.x < d.NW) nws[3 * VC_NWE + threadIdx.x] = b.lat[VC_SITE_NUOMEGA][threadIdx.x];
  }
  // sharded step: the exchange buffer comes back from the sum holding every rank's contribution; the elements no block of
  // this rank writes (PW rows beyond this rank's cell blocks, the header, the nu_omega slots of the gradient region) are
  // cleared here so that they do not re-enter the next sum
  if (phase == VC_PH_A && cblock == 0) {
    for (int tt = d.nb_tail_cell * d.NW + (int)threadIdx.x; tt < xb.pw_cap * d.NW; tt += VC_TC) vc_xput(xb, xb.pw_off + tt, 0.f);
    if (threadIdx.x < 4) vc_xput(xb, threadIdx.x, 0.f);
    if (vel) {
      const bool lrmn = d.guide == VC_GUIDE_LRMN;
      const int fin_per = lrmn ? d.R + 2 : 2;
      for (int tt = threadIdx.x; tt < d.NW * fin_per; tt += VC_TC) vc_xput(xb, vc_nuw_elem_off(d, lrmn, tt / fin_per, tt % fin_per), 0.f);
    }
  }
  VC_WSTAMP(0, 4);
  // ---- phi_xy sample of step s, phase, Fourier basis, cell record (omega is filled by K_omega) -------------------
  float s1_new = sc_old.x, c1_new = sc_old.y;      // conditioned phases: the record keeps its sin / cos
  if (in_range) {
    const float px = pxy.x, py = pxy.y;
    float x, y;
    if (cxy) {
      x = b.cnd[VC_SITE_PHIXY][2 * c]; y = b.cnd[VC_SITE_PHIXY][2 * c + 1];
      loss += 0.5 * ((double)(x - px) * (x - px) + (double)(y - py) * (y - py)) + (double)VC_LOG_2PI;
    } else {
      x = pp.x + ex;
      y = pp.y + ey;
      loss += 0.5 * ((double)(x - px) * (x - px) + (double)(y - py) * (y - py)) - 0.5 * ((double)ex * ex + (double)ey * ey);
    }
    if (!cxy || boot) {          // conditioned phases never change: their record is written once
      *reinterpret_cast<float2*>(b.lat[VC_SITE_PHIXY] + 2 * (size_t)c) = make_float2(x, y);
      const float ph = atan2f(y, x);          // (the deterministic site only)
      float s1, c1;
      vc_dir_sincos(x, y, &s1, &c1);
      float sk[VC_MAXH], ck[VC_MAXH];
      sk[0] = s1; ck[0] = c1;
      const int hm = d.H > d.Hw ? d.H : d.Hw;          // the W table (vc_put_w) goes up to Hw
      for (int k = 1; k < hm && k < VC_MAXH; ++k) {
        sk[k] = sk[k - 1] * c1 + ck[k - 1] * s1;
        ck[k] = ck[k - 1] * c1 - sk[k - 1] * s1;
      }
      s1_new = s1; c1_new = c1;
      float2* ct = reinterpret_cast<float2*>(b.CT + (size_t)cp * d.ctw);
      for (int k = 0; k < d.H; ++k) { ct[2 * k] = make_float2(sk[k], sk[k]); ct[2 * k + 1] = make_float2(ck[k], ck[k]); }
      if (vel) vc_put_w(d, b, c, cp, sk, ck);      // the W row of K_main's nu_omega partials follows the phase
      if (boot) {              // step-invariant entries of the record
        for (int q = 0; q < d.nbk; ++q) {
          const float v = b.Dbm[(size_t)q * d.Nc + c];
          ct[2 * d.H + q] = make_float2(v, v);
        }
        const int nbk = d.nbk;
        if (!vel) ct[2 * d.H + nbk] = make_float2(0.f, 0.f);
        { const float cfs = b.cf[c] * vc_rec_cf_scale(d.noise); ct[2 * d.H + nbk + 1] = make_float2(cfs, cfs); }
        if (!vel) { b.lat_omega[c] = 0.f; b.lat_domega[c] = 0.f; }
      }
      b.lat_phi[c] = ph;
    }
  }
  VC_WSTAMP(0, 5);
  if (OMEGA && vel) {
    // the nu_omega chain (gradient from K_main's partials, optimiser, next sample) and omega_c of this block's cells
    if (cwm) {
      __syncthreads();        // the chain wave has left nu_omega of step s in sh_nuw
      vc_nuw_cells(d, b, in_range ? c : d.Nc, s1_new, c1_new, sh_nuw);
    } else
      vc_nuw_chain(d, b, P, G, s, seed, VcAdamArgs{Mm, Vv, 0.0, 0.0, 0.0, 0.0, o.b1, o.b2, o.eps, o.clip, header, o.wd, 0, o.frozen}, boot, cblock == 0, phase,
                   xb, in_range ? c : d.Nc, s1_new, c1_new, VC_TC, sh_nuw);
  }
  {
    const double ws = vc_wave_sum_d63(loss);
    if (lane == 63) sm_lc[wave] = ws;
    __syncthreads();                        // (only the live waves take part; also orders sm_w)
    if (vel && !boot && !(phase == VC_PH_ALL && d.pw_inline) && (int)threadIdx.x < d.NW) {
      const int j = threadIdx.x;
      float t = 0.f;
      for (int w = 0; w < VC_TC / 64; ++w) t += sm_w[w][j];
      if (phase == VC_PH_A) vc_xput(xb, xb.pw_off + cblock * d.NW + j, t);
      else b.PW[(size_t)cblock * d.NW + j] = t;
    }
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < VC_TC / 64; ++w) t += sm_lc[w];
      b.LPF[(size_t)(s & 1) * d.nlpf + d.nb_post_gene + cblock] = t;
    }
  }
  VC_WSTAMP(0, 6);
}

// Phase A: the part of this rank's loss that is complete before the launch -- prior / guide terms of the sample of the step
// being finished (LPF, written one step earlier), K_main's likelihood partials, the step-invariant constant -- folded in
// fixed order into one double and handed to the exchange (vc_loss_split)
__device__ __forceinline__ void vc_tail_loss_base_block(const VcDims& d, const VcBufs& b, long long s, const VcXb xb) {
  __shared__ double sm_lb[16];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  double sl = 0.0;
  const double* lpf = b.LPF + (size_t)((s - 1) & 1) * d.nlpf;
  for (int i = t; i < d.nlpf; i += 1024) sl += lpf[i];
  for (int i = t; i < d.n_main_wg; i += 1024) sl -= (double)b.LO[i];
  sl = vc_wave_sum_d63(sl);
  if (lane == 63) sm_lb[wv] = sl;
  __syncthreads();
  if (t == 0) {
    double tot = b.const_loss;
    for (int w = 0; w < 16; ++w) tot += sm_lb[w];
    vc_loss_split(tot, xb.x + xb.loss_off, xb.xmode);
  }
}

template <int MQ, int phase, int SPEC = 0>
__global__ __launch_bounds__(1024) void vc_tail_kernel(const VcDims d, const VcBufs b, float* __restrict__ P,
                                                       float* __restrict__ G, const long long* __restrict__ step_dev,
                                                       uint64_t seed, const VcAdamArgs a, int boot, const VcXb xb) {
  vc_spec_assume<SPEC>(d);
  if (SPEC > 0) __builtin_assume(boot == 0);      // (vc_launch_tail: the compiled signatures serve the steady-state launches only)
  // s: index of the step whose sample this launch draws (boot: the step about to run; else K_main has advanced the
  // counter, s = t + 1 is also the 1-based optimiser step of the update applied here)
  const long long s = *step_dev;
  VcOpt o;
  o.step_size = boot ? 0.f : b.step_size[0];       // written by K_main together with the counter
  o.b1 = a.b1; o.b2 = a.b2; o.eps = a.eps; o.clip = a.clip; o.c2 = b.step_size[1]; o.wd = a.wd; o.frozen = a.frozen;
  if ((int)blockIdx.x < d.nb_post_gene) vc_tail_gene_block<MQ, phase>(d, b, P, G, a.m, a.v, a.header, blockIdx.x, s, seed, o, boot, xb);
  else if ((int)blockIdx.x < d.nb_post_gene + d.nb_tail_cell)
    vc_tail_cell_block<phase>(d, b, P, G, a.m, a.v, a.header, blockIdx.x - d.nb_post_gene, s, seed, o, boot, xb);
  else if (phase == VC_PH_A) vc_tail_loss_base_block(d, b, s, xb);
}

void vc_launch_tail(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                    const VcAdamArgs& a, int boot, int phase, const VcXb& xb, hipStream_t st) {
  // phase A: + the loss-base block (phase B has its own launch: vc_launch_phase_b)
  const int nblk = d.nb_post_gene + d.nb_tail_cell + (phase == VC_PH_A ? 1 : 0);
  const dim3 grid(nblk), block(1024);
  // phase A of a configuration with a compiled signature (vc_tail_spec.h)
  if (phase == VC_PH_A && !boot && vc_spec_launch<VC_SPECK_SHARDED, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_tail_kernel<decltype(mq)::value, VC_PH_A, decltype(sp)::value>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, boot, xb);
      }))
    return;
  // the phase is a template parameter: the single-rank kernel carries none of the sharded step's code (or registers)
#define VC_TAIL_LAUNCH(MQ_)                                                                                                       \
  do {                                                                                                                            \
    if (phase == VC_PH_A) hipLaunchKernelGGL((vc_tail_kernel<MQ_, VC_PH_A>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, boot, xb); \
    else hipLaunchKernelGGL((vc_tail_kernel<MQ_, VC_PH_ALL>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, boot, xb); \
  } while (0)
  if (d.nq <= 2) VC_TAIL_LAUNCH(2);
  else if (d.nq <= 4) VC_TAIL_LAUNCH(4);
  else if (d.nq <= 6) VC_TAIL_LAUNCH(6);
  else VC_TAIL_LAUNCH(VC_MAXQ);
#undef VC_TAIL_LAUNCH
}

// ---------------------------------------------------------------------------------------------
// K_omega: blocks [0, nb_cell): nu_omega (gradient, optimiser, next sample) redundantly per block, then omega_c into the
// cell records of the block's 256 cells -- the only chain the next K_main waits for.  Off that chain, in blocks of their
// own: the loss of the finished step (one block), the negative-binomial histogram terms for shape_inv of the next step
// (one wave per task), and the Philox draws of the step after next into the eps ring.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void vc_omega_loss_block(const VcDims& d, const VcBufs& b, float* __restrict__ G, long long s,
                                                    double* __restrict__ loss_dev, long long loss_slots, int phase,
                                                    const VcXb xb) {
  __shared__ double sm_lossw[4];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  double sl = 0.0;
  if (phase == VC_PH_B) {
    // every rank's loss terms, summed piece by piece by the exchange: base + one per gene block, fixed order
    for (int i = t; i < 1 + d.nb_post_gene; i += 256) sl += vc_loss_join_x(xb, xb.loss_off + VC_LOSS_PIECES * i);
  } else {
    // everything below belongs to the sample of the finished step s - 1 and was complete BEFORE this launch: prior / guide
    // terms and the r-only likelihood term (written when the sample was drawn, half (s - 1) & 1), its histogram sums (half
    // (s - 1) & 1 when shape_inv is learned), K_main's likelihood partials -- no block of this launch writes any of it
    const double* lpf = b.LPF + (size_t)((s - 1) & 1) * d.nlpf;
#pragma unroll 4
    for (int i = t; i < d.nlpf; i += 256) sl += lpf[i];
    const double* lpr = b.LPR + (size_t)((s - 1) & 1) * d.nb_post_gene;
#pragma unroll 4
    for (int i = t; i < d.nb_post_gene; i += 256) sl += lpr[i];
    if (d.nmat_r > 0) {
      // the one LONG list (a term per histogram task: 16 per thread at 2 000 genes x 2 matrices) in trips of eight REQUESTS, then the
      // eight adds in index order: as `sl -= hl[i]` hipcc waits for every single load -- 16 dependent round trips, the block ended with
      // the gene blocks at 7 us (round 6, found next to the histogram prefetch).  The short lists stay plain loops: in trips they cost
      // the tutorial flow's merged tail 0.75 us per step (measured, same box: 72.35 vs 71.6 us).
      const double* hl = b.HL + (size_t)(d.hist_par ? ((s - 1) & 1) : 0) * b.n_tasks;
      const int n = b.n_tasks;
      for (int i0 = t; i0 < n; i0 += 8 * 256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; v[u] = hl[i < n ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + 256 * u < n) sl -= v[u];
      }
    }
#pragma unroll 4
    for (int i = t; i < d.n_main_wg; i += 256) sl -= (double)b.LO[i];
  }
  sl = vc_wave_sum_d63(sl);
  if (lane == 63) sm_lossw[wv] = sl;
  __syncthreads();
  if (t == 0) {
    const double loss = ((sm_lossw[0] + sm_lossw[1]) + (sm_lossw[2] + sm_lossw[3])) + (phase == VC_PH_B ? 0.0 : b.const_loss);
    const long long step = s - 1;
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
    VC_WSTAMP(1, 0);
  }
}

// Histogram terms of shape_inv(s) for the 16 tasks of one 1024-thread block while the gene blocks of the SAME launch are still
// computing shape_inv(s): the update is re-derived here from the snapshot the launch before took (SIS, half (s - 1) & 1),
// K_main's partials and the histogram sums of the finished step -- statement by statement what the owning (gene, role 12) thread
// runs, the second-stage reduction in its order included: the same bits.  The 16 tasks of a block belong to a handful of
// consecutive genes; wave v adds, for the gene of every task, the chunks v, v + 16, ... (what wave v of that gene's gene block
// adds), lane i for task i; the 16 wave sums meet in the LDS and every wave adds them in wave order for its own task's gene.
// K_main's partial row of shape_inv is read about once this way (a wave per task on its own read it sixteen times over: 34 MB).
__device__ __forceinline__ void vc_hist_rederive_block(const VcDims& d, const VcBufs& b, const float* __restrict__ P, long long s,
                                                       const VcAdamArgs& a, int task0) {
  __shared__ float sm_a[VC_PG_WAVES][VC_PG_WAVES];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t NP = d.Ng_pad;
  const int q = d.kind == VC_KIND_PHASE ? d.Kq : d.Kq + 2;
  const int task = task0 + wv;
  const bool have = task < b.n_tasks;
  // everything that depends on the task table only, requested together
  const int ti = task0 + (lane < VC_PG_WAVES ? lane : 0);
  const int gi = b.h_task[4 * (ti < b.n_tasks ? ti : b.n_tasks - 1)];          // lane i < 16: the gene of task i of this block
  const int g = b.h_task[4 * (have ? task : b.n_tasks - 1)];                   // this wave's own gene
  float acc = 0.f;
  if (lane < VC_PG_WAVES && d.kind != VC_KIND_VU) {
    // eight chunks requested per trip, added in chunk order (a plain `acc += load` loop is eight dependent round trips)
    constexpr int UB = 8;
    const VcChunkWalk wk = vc_chunk_walk_lane(d, b, gi, wv);       // (lane i: the gene of task i -- its own gene block's batch ranges)
    for (int ch0 = wk.first; ch0 < wk.end; ch0 += UB * wk.stride) {
      float v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int ch = ch0 + u * wk.stride;
        v[u] = b.GO[((size_t)(ch < wk.end ? ch : ch0) * d.nq + q) * NP + gi];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (ch0 + u * wk.stride < wk.end) acc += v[u];
    }
  }
  const float* sis = b.SIS + (size_t)((s - 1) & 1) * 4 * NP + g;
  const float p0 = sis[0], si = sis[3 * NP];
  float mm = sis[NP], vv = sis[2 * NP];
  double HDg = 0.0;
  {
    const double* __restrict__ HDs = b.HD + (size_t)((s - 1) & 1) * b.n_tasks;
    const int t0 = b.h_tptr[g], t1 = b.h_tptr[g + 1];
    for (int tb = t0; tb < t1; tb += 4) {
      double hd[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) hd[k] = HDs[tb + k < t1 ? tb + k : tb];
#pragma unroll
      for (int k = 0; k < 4; ++k) if (tb + k < t1) HDg += hd[k];
    }
  }
  if (lane < VC_PG_WAVES) sm_a[wv][lane] = acc;
  __syncthreads();
  float U_r = 0.f;
#pragma unroll
  for (int w = 0; w < VC_PG_WAVES; ++w) U_r += sm_a[w][wv];          // wave order, as the gene block's T()
  if (d.kind == VC_KIND_VU) U_r = 0.f;
  const float gg = vc_si_grad(d, 1.0f / si, si, U_r, HDg, d.root_w);
  const float np = vc_adam_elem(p0, gg, mm, vv, b.step_size[0], a.b1, a.b2, a.eps, a.clip, b.step_size[1],
                                vc_wd_at(a.wd, a.frozen, d.poff[VC_P_SHAPE_INV_ULOCS] + g));
  VC_WSTAMP(1, 6);
  if (have) vc_hist_wave(d, b, P, 0, task, lane, expf(np), (int)(s & 1));
  VC_WSTAMP(1, 7);
}

// The same with the dense tables (d.hist_dense): ONE block per gene block -- lane = gene, wave v adds the chunks v, v + 16, ... of
// K_main's shape_inv row exactly as wave v of the gene block does (coalesced rows), the 16 wave sums meet in the LDS, every wave
// then holds shape_inv(s) of the block's 64 genes and takes its slices of the count axis (vc_hist_dense_block).
// QT: a quarter block (vc_hist_dense16_finish<true>) -- four live waves, a thread = (gene of the quarter, virtual wave v of the gene
// block that owns the gene): it adds the chunks wave v of the gene block adds, in that order; waves 4 .. 15 leave at once.
template <bool QT>
__device__ __forceinline__ void vc_hist_rederive_dense(const VcDims& d, const VcBufs& b, long long s, const VcAdamArgs& a, int gb,
                                                       double* sm_hd, int msel, int quarter) {
  __shared__ float sm_a[VC_PG_WAVES][64];
  if (QT && threadIdx.x >= 256) return;
  const int lane = threadIdx.x & 63, col = QT ? (lane & 15) : lane;
  const size_t NP = d.Ng_pad;
  const int q = d.kind == VC_KIND_PHASE ? d.Kq : d.Kq + 2;
  VcHistPre hp;
  vc_hist_dense16_rows<QT>(d, b, gb, hp, msel, quarter);    // (arrives with the loads below)
  const int g = gb * 64 + hp.gi;
  const bool live = g < d.Ng;
  float acc = 0.f;
  if (d.kind != VC_KIND_VU) {
    constexpr int UB = 8;
    const VcChunkWalk wk = vc_chunk_walk(d, b, gb * 64, hp.v);       // as wave hp.v of the gene block that owns these 64 genes
    for (int ch0 = wk.first; QT ? __builtin_amdgcn_ballot_w64(ch0 < wk.end) != 0ull : ch0 < wk.end; ch0 += UB * wk.stride) {
      float v[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int ch = ch0 + u * wk.stride;
        // (a chunk that is not this thread's: a row that exists -- the value is dropped)
        v[u] = b.GO[((size_t)(ch < wk.end ? ch : (QT ? 0 : ch0)) * d.nq + q) * NP + g];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (ch0 + u * wk.stride < wk.end) acc += v[u];
    }
  }
  VC_WSTAMP(1, 2);
  vc_hist_dense16_issue<QT>(d, b, hp);         // the table rows travel while the update is re-derived
  const float* sis = b.SIS + (size_t)((s - 1) & 1) * 4 * NP + g;
  const float p0 = sis[0], si = live ? sis[3 * NP] : 1.f;
  float mm = sis[NP], vv = sis[2 * NP];
  double HDg = 0.0;
  if (live) {
    const double* __restrict__ HDs = b.HD + (size_t)((s - 1) & 1) * b.n_tasks;
    const int t0 = b.h_tptr[g], t1 = b.h_tptr[g + 1];
    for (int tb = t0; tb < t1; tb += 4) {
      double hd[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) hd[k] = HDs[tb + k < t1 ? tb + k : tb];
#pragma unroll
      for (int k = 0; k < 4; ++k) if (tb + k < t1) HDg += hd[k];
    }
  }
  VC_WSTAMP(1, 4);
  sm_a[hp.v][col] = acc;
  __syncthreads();
  float U_r = 0.f;
#pragma unroll
  for (int w = 0; w < VC_PG_WAVES; ++w) U_r += sm_a[w][col];          // wave order, as the gene block's T()
  if (d.kind == VC_KIND_VU) U_r = 0.f;
  const float gg = vc_si_grad(d, 1.0f / si, si, U_r, HDg, d.root_w);
  const float np = vc_adam_elem(p0, gg, mm, vv, b.step_size[0], a.b1, a.b2, a.eps, a.clip, b.step_size[1],
                                vc_wd_at(a.wd, a.frozen, d.poff[VC_P_SHAPE_INV_ULOCS] + (live ? g : 0)));
  VC_WSTAMP(1, 6);
  vc_hist_dense16_finish<QT>(d, b, gb, live ? expf(np) : 1.f, (int)(s & 1), hp, sm_hd);
  VC_WSTAMP(1, 7);
}

// The blocks of K_omega's grid that are off the nu_omega chain (256 threads; `xblk` = index behind the cell blocks): 0 = the loss
// of the finished step, then nb_hist histogram blocks (4 tasks each), then the eps blocks.  rederive: the gene blocks of the
// same launch are computing shape_inv(s) right now (vc_tail2_kernel)
template <bool TAIL2>
__device__ __forceinline__ void vc_omega_extra_block(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                                     const long long s, uint64_t seed, const VcAdamArgs& a,
                                                     double* __restrict__ loss_dev, long long loss_slots, int boot, int nb_hist,
                                                     int xblk, int phase, const VcXb xb, bool rederive, int nthr = 256) {
  // nthr = 256: K_omega's own grid (4 tasks / 256 eps pairs per block); 1024: a launch of 1024-thread blocks (vc_tail2_kernel) --
  // a block is placed as a whole 16-wave slot there (2 per CU), so every wave of it takes a task / 64 pairs: a quarter of the
  // blocks, or the extras alone would need three more rounds of slots than the chains they run beside
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  if (xblk == 0) {                                 // the loss of the finished step
    if (t >= 256) return;                          // (a 256-thread job: the other waves leave before its barrier)
    if (!boot) vc_omega_loss_block(d, b, G, s, loss_dev, loss_slots, phase, xb);
    return;
  }
  xblk -= 1;
  if (xblk < nb_hist) {                            // histogram terms of shape_inv(s)
    const int half = d.hist_par ? (int)(s & 1) : 0;
    if (d.hist_dense) {                            // dense tables: one block per gene block (barriers inside)
      double* sm_hd = vc_hist_lds();
      if (TAIL2) {                                 // (compiled into the one-launch tail only: 16-wave blocks, update re-derived)
        // vc_launch_tail2 asks for 4 x b.n_hc_split + nm x Ng_pad / 64 blocks: first the quarter blocks of the gene blocks that hold a
        // highly expressed gene (the longest evaluations of the launch), then one block per (matrix, gene block) -- of which those leave
        const int nblk = d.Ng_pad / 64, nm = d.model == VC_MODEL_VELOCITY ? 2 : 1;
        if (xblk < 4 * b.n_hc_split) {
          typedef const __attribute__((address_space(4))) int* ciptr;
          const int id = ((ciptr)(const void*)b.hc_split)[xblk >> 2], quarter = xblk & 3;
          const int msel = id / nblk, gb = id % nblk;
          if (rederive && !CND(VC_SITE_SHAPE_INV)) { vc_hist_rederive_dense<true>(d, b, s, a, gb, sm_hd, msel, quarter); return; }
          if (t >= 256) return;                    // (four live waves: 16 genes x 16 slices)
          VcHistPre hp;
          vc_hist_dense16_rows<true>(d, b, gb, hp, msel, quarter);
          vc_hist_dense16_issue<true>(d, b, hp);
          vc_hist_dense16_finish<true>(d, b, gb, vc_hist_si(d, b, P, 0, gb * 64 + hp.gi), half, hp, sm_hd);
          return;
        }
        xblk -= 4 * b.n_hc_split;
        const int msel = nm > 1 ? xblk / nblk : 0, gb = nm > 1 ? xblk % nblk : xblk;
        if (b.hc_rows[msel * nblk + gb] > VC_HIST_SPLIT_ROWS) return;       // (its quarter blocks do the work)
        if (rederive && !CND(VC_SITE_SHAPE_INV)) { vc_hist_rederive_dense<false>(d, b, s, a, gb, sm_hd, msel, 0); return; }
        VcHistPre hp;
        vc_hist_dense16_rows<false>(d, b, gb, hp, msel, 0);
        vc_hist_dense16_issue<false>(d, b, hp);
        vc_hist_dense16_finish<false>(d, b, gb, vc_hist_si(d, b, P, 0, gb * 64 + lane), half, hp, sm_hd);
        return;
      }
      const int g = xblk * 64 + lane;
      float si = vc_hist_si(d, b, P, 0, g);
      if (phase == VC_PH_B && !CND(VC_SITE_SHAPE_INV) && g < d.Ng) {
        // the gene blocks of this launch are rewriting shape_inv: re-derive its update from phase A's snapshot and the summed
        // gradient (vc_adam_elem is the arithmetic the owning thread runs: the same bits)
        const long long off = d.poff[VC_P_SHAPE_INV_ULOCS] + g;
        float mm = vc_xsis(xb, d.Ng_pad + g), vv = vc_xsis(xb, 2 * (size_t)d.Ng_pad + g);
        si = expf(vc_adam_elem(vc_xsis(xb, g), vc_xget(xb, off), mm, vv, b.step_size[0], a.b1, a.b2, a.eps, a.clip, b.step_size[1], vc_wd_at(a.wd, a.frozen, off)));
      }
      vc_hist_dense_block(d, b, xblk, si, half, nthr >> 6, sm_hd);
      return;
    }
    const int task = xblk * (nthr >> 6) + wv;      // lists of distinct values: one wave per task
    if (TAIL2 && rederive && !CND(VC_SITE_SHAPE_INV)) {           // (block-wide: barriers inside; nthr == 1024)
      // two rounds of 16 tasks per block: half the blocks to place (every 1024-thread block of this launch holds a CU on its own)
#pragma unroll 1
      for (int rnd = 0; rnd < VC_HIST_ROUNDS; ++rnd) {
        vc_hist_rederive_block(d, b, P, s, a, (xblk * VC_HIST_ROUNDS + rnd) * (nthr >> 6));
        __syncthreads();                                   // the LDS rows of the next round overwrite this one's
      }
      return;
    }
    if (task < b.n_tasks) {
      if (phase == VC_PH_B && !CND(VC_SITE_SHAPE_INV)) {
        // the gene blocks of this launch are rewriting shape_inv: re-derive its update from phase A's snapshot and the
        // summed gradient (vc_adam_elem is the arithmetic the owning thread runs: the same bits)
        const int g = b.h_task[4 * task];
        const long long off = d.poff[VC_P_SHAPE_INV_ULOCS] + g;
        float mm = vc_xsis(xb, d.Ng_pad + g), vv = vc_xsis(xb, 2 * (size_t)d.Ng_pad + g);
        const float np = vc_adam_elem(vc_xsis(xb, g), vc_xget(xb, off), mm, vv, b.step_size[0], a.b1, a.b2, a.eps, a.clip, b.step_size[1], vc_wd_at(a.wd, a.frozen, off));
        vc_hist_wave(d, b, P, 0, task, lane, expf(np), half);
      } else {
        vc_hist_wave(d, b, P, 0, task, lane, -1.f, half);
      }
    }
    return;
  }
  xblk -= nb_hist;
  // eps ring: the draws of step s + 1 (and, when booting, of step s) -- one Philox block = two consecutive indices.
  // Three slots: this launch reads the slots of s - 1 and s and writes the slot of s + 1.  The ring holds this rank's slice
  // of the stream: replicated sites at their global index, phi_xy shifted by the rank's first cell.
  const int per = nthr == 1024 ? VC_EPS_PER_THREAD : 1;          // pairs per thread (1024-thread launches: fewer, fatter blocks)
  for (int k = 0; k < per; ++k) {
    const long long pair = ((long long)xblk * per + k) * nthr + t;
    if (2 * pair < d.eps_total) {
      const uint64_t gpair = (uint64_t)(pair + (2 * pair >= d.eps_n_global ? d.cell_offset : 0));
      float n0, n1;
      vc_philox_normal2(seed, s + 1, gpair, n0, n1);
      *reinterpret_cast<float2*>(b.EPS + (size_t)((s + 1) % 3) * d.eps_total + 2 * pair) = make_float2(n0, n1);
      if (boot) {
        vc_philox_normal2(seed, s, gpair, n0, n1);
        *reinterpret_cast<float2*>(b.EPS + (size_t)(s % 3) * d.eps_total + 2 * pair) = make_float2(n0, n1);
      }
    }
  }
  VC_WSTAMP(1, 3);
}

// The nu_omega chain for a block of `nthr` threads (256: K_omega's blocks; the cell-block size of the one-launch tail): per-coefficient
// sums of the partials of d loglik / d nu_omega (redundantly in every block, fixed order) -> gradient + ClippedAdam of the
// nu_omega-related parameters on a snapshot (`first`: this block stores them and the snapshot of the next step) -> the nu_omega
// sample of step s -> omega_c and d omega / d phi of cell c (sin, cos of its phase of step s: s1, c1) into its record.
// part 1 of the chain: per-coefficient sums of the partials of d loglik / d nu_omega into sh.up -- every block, fixed order (lane i
// adds rows i, i + 64, ... in double, then the 64 lanes: independent of the number of waves that share the coefficients).
// In two halves, so that a caller can put its own loads between them: `issue` only REQUESTS the rows of this wave's first two
// coefficients (registers, nothing consumed), `finish` adds them up.
__device__ __forceinline__ void vc_nuw_sums_issue(const VcDims& d, const VcBufs& b, int boot, int phase, const VcXb& xb, int nthr,
                                                  VcNuwRaw& raw) {
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, nwv = nthr >> 6;
  const bool pwm = phase != VC_PH_B && d.pw_inline;           // single rank: K_main's own partials
  const float* __restrict__ PWs = phase == VC_PH_B ? xb.x + xb.pw_off : (pwm ? b.PWM : b.PW);
  const int n_pw = phase == VC_PH_B ? xb.pw_cap : (pwm ? d.n_main_wg : d.nb_tail_cell);
  const int pw_ld = pwm ? d.pw_inline : d.NW;
  const bool on = !boot && n_pw <= 64 * VC_NUW_RAW;      // (uniform)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = wv + nwv * q, jc = j < d.NW ? j : 0;
    // no branch around a load and no select behind it: a clamped row, and `finish` adds only the rows that exist, for coefficients that
    // exist (with the select here hipcc waited for every single load: 11 dependent round trips on the chain wave; round 6)
    if (on) {
#pragma unroll
      for (int k = 0; k < VC_NUW_RAW; ++k) {
        const int i = lane + 64 * k, ic = i < n_pw ? i : 0;
        raw.r[q][k] = PWs[(size_t)ic * pw_ld + jc];
      }
    } else {
#pragma unroll
      for (int k = 0; k < VC_NUW_RAW; ++k) raw.r[q][k] = 0.f;
    }
  }
}
__device__ __forceinline__ void vc_nuw_sums_finish(const VcDims& d, const VcBufs& b, int boot, int phase, const VcXb& xb, int nthr,
                                                   const VcNuwRaw& raw, VcNuwShared& sh) {
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, nwv = nthr >> 6;
  // phase B: the PW rows are the sums over ranks (every rank holds the complete gradient: prior / entropy weight 1)
  const bool pwm = phase != VC_PH_B && d.pw_inline;           // single rank: K_main's own partials
  const float* __restrict__ PWs = phase == VC_PH_B ? xb.x + xb.pw_off : (pwm ? b.PWM : b.PW);
  const int n_pw = phase == VC_PH_B ? xb.pw_cap : (pwm ? d.n_main_wg : d.nb_tail_cell);
  const int pw_ld = pwm ? d.pw_inline : d.NW;
  const int nw = d.NW;
  VC_WSTAMP(1, 1);
  if (!boot) {
    double u[2] = {0.0, 0.0};
    if (n_pw <= 64 * VC_NUW_RAW) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < VC_NUW_RAW; ++k)
          if (lane + 64 * k < n_pw) u[q] += (double)raw.r[q][k];          // rows lane, lane + 64, ... in this order
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = wv + nwv * q;
        if (j < nw)
          u[q] = vc_col_sum_d(PWs, n_pw, pw_ld, j, lane, u[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = wv + nwv * q;
      if (j < nw) { const double r = vc_wave_sum_d63(u[q]); if (lane == 63) sh.up[j] = (float)r; }
    }
    for (int j = wv + 2 * nwv; j < nw; j += nwv) {
      double r = 0.0;
      r = vc_col_sum_d(PWs, n_pw, pw_ld, j, lane, r);
      r = vc_wave_sum_d63(r);
      if (lane == 63) sh.up[j] = (float)r;
    }
  }
}

// the same sums in one piece (K_omega's own blocks, phase B: nothing to put between request and use, and no 24 registers held
// for it -- phase B's kernel stays below 64 VGPRs, two blocks per CU)
__device__ __forceinline__ void vc_nuw_sums_direct(const VcDims& d, const VcBufs& b, int boot, int phase, const VcXb& xb, int nthr,
                                                   VcNuwShared& sh) {
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, nwv = nthr >> 6;
  const bool pwm = phase != VC_PH_B && d.pw_inline;
  const float* __restrict__ PWs = phase == VC_PH_B ? xb.x + xb.pw_off : (pwm ? b.PWM : b.PW);
  const int n_pw = phase == VC_PH_B ? xb.pw_cap : (pwm ? d.n_main_wg : d.nb_tail_cell);
  const int pw_ld = pwm ? d.pw_inline : d.NW;
  const int nw = d.NW;
  VC_WSTAMP(1, 1);
  if (!boot && phase == VC_PH_B && xb.nslots > 0) {
    // the exchange folded into this launch: every row is the sum of the ranks' rows, added here in rank order (vc_xget) -- the same
    // association as below, row by row
    for (int j = wv; j < nw; j += nwv) {
      double r = 0.0;
      for (int i = lane; i < n_pw; i += 64) r += (double)vc_xget(xb, xb.pw_off + (long long)i * pw_ld + j);
      r = vc_wave_sum_d63(r);
      if (lane == 63) sh.up[j] = (float)r;
    }
    return;
  }
  if (!boot) {
    double u[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = wv + nwv * q;
      if (j < nw)
        u[q] = vc_col_sum_d(PWs, n_pw, pw_ld, j, lane, u[q]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = wv + nwv * q;
      if (j < nw) { const double r = vc_wave_sum_d63(u[q]); if (lane == 63) sh.up[j] = (float)r; }
    }
    for (int j = wv + 2 * nwv; j < nw; j += nwv) {
      double r = 0.0;
      r = vc_col_sum_d(PWs, n_pw, pw_ld, j, lane, r);
      r = vc_wave_sum_d63(r);
      if (lane == 63) sh.up[j] = (float)r;
    }
  }
}

// part 2 (sh.up filled by every wave that took part in part 1; the barrier below orders it): gradient, optimiser and next sample
// of the nu_omega-related parameters.  WAVE: ONE wave runs it (the chain wave of vc_tail_cell_block: its 64 lanes stride the
// elements, the block barriers become wavefront-scope fences) -- the same statements per element, the same bits
template <bool WAVE>
__device__ __forceinline__ void vc_nuw_params(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                              long long s, uint64_t seed, const VcAdamArgs& a, int boot, bool first, int phase,
                                              int nthr_, VcNuwShared& sh) {
  const int t = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
  const int nthr = WAVE ? 64 : nthr_;
  auto sync = [&]() {
    if (WAVE) {           // LDS operations of one wave complete in order; the fence keeps the compiler from moving them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else __syncthreads();
  };
  const bool lrmn = d.guide == VC_GUIDE_LRMN;
  const float rw = phase == VC_PH_B ? 1.f : d.root_w;
  const int nw = d.NW;
  const int fin_per = lrmn ? d.R + 2 : 2;
  const int nelem = nw * fin_per;
  const bool cnd = CND(VC_SITE_NUOMEGA);
  // eps index of element (j, ce): mean-field ce = 1: nu_omega[j]; LRMN ce = 1..R: eps_W[ce - 1], ce = R + 1: eps_D[Ng + j]
  auto eps_index = [&](int j, int ce) -> long long {
    if (!lrmn) return ce == 1 ? d.eoff[VC_E_NUOMEGA] + j : -1;
    if (ce == 0) return -1;
    return ce <= d.R ? d.eoff[VC_E_LRMN_W] + (ce - 1) : d.eoff[VC_E_LRMN_D] + (long long)d.Ng + j;
  };
  const float* __restrict__ eps_new = b.EPS + (size_t)(s % 3) * d.eps_total;
  const float* __restrict__ eps_old = b.EPS + (size_t)((s + 2) % 3) * d.eps_total;
  const float* __restrict__ nws = b.NWS + (size_t)(s & 1) * 4 * VC_NWE;        // the nu_omega snapshot of this step ...
  float* __restrict__ nws_next = b.NWS + (size_t)((s + 1) & 1) * 4 * VC_NWE;   // ... and the copy block 0 fills for the next
  sync();
  VC_WSTAMP(1, 2);
  // ---- gradient + ClippedAdam of the nu_omega-related parameters: one thread per element, every block alike ----------
  const float step_size = boot ? 0.f : b.step_size[0];
  for (int tt = t; tt < nelem; tt += nthr) {
    const int j = tt / fin_per, ce = tt % fin_per;
    const long long off = vc_nuw_elem_off(d, lrmn, j, ce);
    float p = nws[tt];                         // the snapshot of this step: block 0 overwrites P / m / v below
    if (!boot) {
      float gx = 0.f;
      if (!cnd) {
        const float x = nws[3 * VC_NWE + j], sd = b.sd_w[j];
        gx = sh.up[j] - rw * (x - b.mu_w[j]) / (sd * sd);
      }
      const long long ei = eps_index(j, ce);
      const float eo = ei >= 0 ? eps_old[ei] : 0.f;          // eps of the finished step for this element
      float gv;
      if (!lrmn) {
        if (ce == 0) gv = -gx;
        else gv = cnd ? 0.f : -gx * expf(p) * eo - rw;
      } else {
        if (ce == 0) gv = -gx;
        else if (ce <= d.R) {
          const float w = expf(p);
          gv = (w > 0.f) ? -gx * eo * w : 0.f;
        } else {
          const float dg = expf(p);
          gv = -gx * eo / (2.f * sqrtf(dg)) * dg;
        }
      }
      float mm = nws[VC_NWE + tt], vv = nws[2 * VC_NWE + tt];
      p = vc_adam_elem(p, gv, mm, vv, step_size, a.b1, a.b2, a.eps, a.clip, b.step_size[1], vc_wd_at(a.wd, a.frozen, off));
      if (first) {
        G[off] = gv; a.m[off - a.header] = mm; a.v[off - a.header] = vv; P[off] = p;
        // ... and the snapshot of the NEXT step (the other copy: nobody reads it in this launch), for a step whose K_tail
        // runs no cell block that would take it (vc_launch_tail_merged, vc_launch_tail2)
        nws_next[tt] = p; nws_next[VC_NWE + tt] = mm; nws_next[2 * VC_NWE + tt] = vv;
      }
    } else if (first) {
      nws_next[tt] = p; nws_next[VC_NWE + tt] = nws[VC_NWE + tt]; nws_next[2 * VC_NWE + tt] = nws[2 * VC_NWE + tt];
    }
    sh.np[tt] = p;
  }
  sync();
  VC_WSTAMP(1, 3);
  // ---- the nu_omega sample of step s ------------------------------------------------------------------------------
  if (t < VC_MAX_NW) sh.lq[t] = 0.0;
  for (int j = t; j < nw; j += nthr) {
    const float* np = sh.np + j * fin_per;
    float val, lq = 0.f;
    const long long i = (long long)d.Ng + j;
    auto en = [&](int ce) { const long long ei = eps_index(j, ce); return boot ? vc_philox_normal(seed, s, ei) : eps_new[ei]; };
    if (!lrmn) {
      const float e = en(1);
      const float u = np[1];
      val = np[0] + expf(u) * e;
      lq = -0.5f * e * e - u - 0.5f * VC_LOG_2PI;
    } else {
      float delta = 0.f;
      for (int k = 0; k < d.R; ++k) delta += expf(np[1 + k]) * en(1 + k);
      const float ed = en(d.R + 1);
      delta += sqrtf(expf(np[d.R + 1])) * ed;
      val = np[0] + delta;
      if (first) b.lat_delta[i] = delta;
    }
    const float x = cnd ? b.cnd[VC_SITE_NUOMEGA][j] : val;
    if (first) {
      b.lat[VC_SITE_NUOMEGA][j] = x;
      nws_next[3 * VC_NWE + j] = x;
      const float lp = vc_normal_lp(x, b.mu_w[j], b.sd_w[j]);
      sh.lq[j] = -(double)d.root_w * ((double)lp - ((cnd || lrmn) ? 0.0 : (double)lq));
    }
    sh.nuw[j] = x;
  }
  sync();
  VC_WSTAMP(1, 4);
  if (first && t == 0) {
    double tot = 0.0;
    for (int j = 0; j < nw; ++j) tot += sh.lq[j];
    b.LPF[(size_t)(s & 1) * d.nlpf + d.nlpf - 1] = tot;
  }
}

// part 3 (sh.nuw complete and visible to the caller's threads): omega_c and d omega / d phi of step s into the record of cell c
// (sin, cos of its phase of step s: s1, c1)
__device__ __forceinline__ void vc_nuw_cells(const VcDims& d, const VcBufs& b, int c, float s1, float c1, const VcNuwShared& sh) {
  if (c < d.Nc) {
    // harmonics by the angle-addition recurrence; fully unrolled (compile-time indices keep sk / ck in registers)
    float sk[VC_MAXH], ck[VC_MAXH];
    sk[0] = s1; ck[0] = c1;
#pragma unroll
    for (int k = 1; k < VC_MAXH; ++k) {
      sk[k] = sk[k - 1] * c1 + ck[k - 1] * s1;
      ck[k] = ck[k - 1] * c1 - sk[k - 1] * s1;
    }
    float omega = 0.f, domega = 0.f;
    for (int xq = 0; xq < d.Nx; ++xq) {
      const float* nwp = sh.nuw + xq * d.Nhw;
      float om = nwp[0], dd = 0.f;
#pragma unroll
      for (int k = 0; k < VC_MAXH; ++k)
        if (k < d.Hw) {
          om += nwp[2 * k + 1] * sk[k] + nwp[2 * k + 2] * ck[k];
          dd += (float)(k + 1) * (nwp[2 * k + 1] * ck[k] - nwp[2 * k + 2] * sk[k]);
        }
      const float dx = b.Dm[(size_t)xq * d.Nc + c];
      omega += dx * om;
      domega += dx * dd;
    }
    vc_rec_put_omega(reinterpret_cast<float2*>(b.CT + (size_t)vc_pos(b, c) * d.ctw), d, omega, sk, ck);
    b.lat_omega[c] = omega;
    b.lat_domega[c] = domega;
  }
  VC_WSTAMP(1, 5);
}

__device__ __forceinline__ void vc_nuw_chain(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                             long long s, uint64_t seed, const VcAdamArgs& a, int boot, bool first, int phase,
                                             const VcXb& xb, int c, float s1, float c1, int nthr, VcNuwShared& sh) {
  vc_nuw_params<false>(d, b, P, G, s, seed, a, boot, first, phase, nthr, sh);
  vc_nuw_cells(d, b, c, s1, c1, sh);
}

// The chain wave of the one-launch tail's cell blocks (vc_tail_cell_block, OMEGA): parts 1 and 2 by ONE wave, beside the block's
// cell waves instead of behind them.  Part 1 in the order of vc_nuw_sums_finish (lane i adds rows i, i + 64, ... in double, then the
// 64 lanes); K_main's rows are 4 or 8 floats long (pw_inline): one or two 16-byte loads fetch every coefficient of a row.
__device__ __forceinline__ void vc_nuw_wave(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                            long long s, uint64_t seed, const VcAdamArgs& a, bool first, VcNuwShared& sh) {
  const int lane = threadIdx.x & 63;
  const float* __restrict__ PWs = b.PWM;
  const int n_pw = d.n_main_wg, pw_ld = d.pw_inline, nw = d.NW;
  VC_WSTAMP(1, 1);
  if (n_pw <= 64 * 8) {
    // up to 512 rows (the 8-genes-per-lane kernels' grid): both halves of an 8-float row are requested before anything is added -- two
    // samples (6 coefficients) otherwise paid a second, dependent round trip; every coefficient still adds its rows in the same order
    constexpr int R = 8;
    float4 raw[2][R];
    const bool two = nw > 4;
#pragma unroll
    for (int k = 0; k < R; ++k) {                       // no branch around a load: a clamped row, the value dropped afterwards
      const int i = lane + 64 * k, ic = i < n_pw ? i : 0;
      raw[0][k] = *reinterpret_cast<const float4*>(PWs + (size_t)ic * pw_ld);
      raw[1][k] = *reinterpret_cast<const float4*>(PWs + (size_t)ic * pw_ld + (two ? 4 : 0));
    }
    double u[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (lane + 64 * k < n_pw) {
        u[0] += (double)raw[0][k].x; u[1] += (double)raw[0][k].y; u[2] += (double)raw[0][k].z; u[3] += (double)raw[0][k].w;
        u[4] += (double)raw[1][k].x; u[5] += (double)raw[1][k].y; u[6] += (double)raw[1][k].z; u[7] += (double)raw[1][k].w;
      }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q >= 4 && !two) break;
      const double r = vc_wave_sum_d63(u[q]);
      if (lane == 63 && q < nw) sh.up[q] = (float)r;
    }
  } else
  for (int j0 = 0; j0 < nw; j0 += 4) {
    double u[4] = {0.0, 0.0, 0.0, 0.0};
    if (n_pw <= 64 * VC_NUW_RAW) {
      float4 raw[VC_NUW_RAW];
#pragma unroll
      for (int k = 0; k < VC_NUW_RAW; ++k) {            // no branch around a load: a clamped row, the value dropped afterwards
        const int i = lane + 64 * k, ic = i < n_pw ? i : 0;
        raw[k] = *reinterpret_cast<const float4*>(PWs + (size_t)ic * pw_ld + j0);
      }
#pragma unroll
      for (int k = 0; k < VC_NUW_RAW; ++k)
        if (lane + 64 * k < n_pw) { u[0] += (double)raw[k].x; u[1] += (double)raw[k].y; u[2] += (double)raw[k].z; u[3] += (double)raw[k].w; }
    } else {
      for (int i = lane; i < n_pw; i += 64) {
        const float4 r = *reinterpret_cast<const float4*>(PWs + (size_t)i * pw_ld + j0);
        u[0] += (double)r.x; u[1] += (double)r.y; u[2] += (double)r.z; u[3] += (double)r.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double r = vc_wave_sum_d63(u[q]);
      if (lane == 63 && j0 + q < nw) sh.up[j0 + q] = (float)r;
    }
  }
  vc_nuw_params<true>(d, b, P, G, s, seed, a, 0, first, VC_PH_ALL, 64, sh);
}

// One block of K_omega's grid (256 threads; `oblk` = its index in that grid).  Also called from the 1024-thread launch of
// phase B, where the waves beyond the fourth have left before the first barrier.
__device__ __forceinline__ void vc_omega_block(const VcDims& d, const VcBufs& b, float* __restrict__ P, float* __restrict__ G,
                                               const long long s, uint64_t seed, const VcAdamArgs& a,
                                               double* __restrict__ loss_dev, long long loss_slots, int boot, int nb_cell,
                                               int nb_hist, int oblk, int phase, const VcXb xb) {
  VC_WSTAMP(1, 0);
  if (oblk >= nb_cell) {
    vc_omega_extra_block<false>(d, b, P, G, s, seed, a, loss_dev, loss_slots, boot, nb_hist, oblk - nb_cell, phase, xb, false);
    return;
  }
  if (d.model != VC_MODEL_VELOCITY) return;
  __shared__ VcNuwShared sh_nuw;
  // this block's cells: the basis of the next phase (written by K_tail) is requested now
  const int c = oblk * 256 + (int)threadIdx.x;
  float s1 = 0.f, c1 = 1.f;
  if (c < d.Nc) {
    const float2* ct = reinterpret_cast<const float2*>(b.CT + (size_t)vc_pos(b, c) * d.ctw);
    s1 = ct[0].x; c1 = ct[1].x;
  }
  vc_nuw_sums_direct(d, b, boot, phase, xb, 256, sh_nuw);
  vc_nuw_chain(d, b, P, G, s, seed, a, boot, oblk == 0, phase, xb, c, s1, c1, 256, sh_nuw);
}

__global__ __launch_bounds__(256) void vc_omega_kernel(const VcDims d, const VcBufs b, float* __restrict__ P,
                                                       float* __restrict__ G, const long long* __restrict__ step_dev,
                                                       uint64_t seed, const VcAdamArgs a, double* __restrict__ loss_dev,
                                                       long long loss_slots, int boot, int nb_cell, int nb_hist) {
  vc_omega_block(d, b, P, G, *step_dev, seed, a, loss_dev, loss_slots, boot, nb_cell, nb_hist, blockIdx.x, VC_PH_ALL, VcXb{});
}

// Phase B of the sharded step in ONE launch: gene blocks (optimiser on the summed gradient + next sample) and K_omega's
// blocks (which depend on phase A and on the exchange, not on the gene blocks) side by side
template <int MQ, int SPEC = 0>
__global__ __launch_bounds__(1024) void vc_phase_b_kernel(const VcDims d, const VcBufs b, float* __restrict__ P,
                                                          float* __restrict__ G, const long long* __restrict__ step_dev,
                                                          uint64_t seed, const VcAdamArgs a, double* __restrict__ loss_dev,
                                                          long long loss_slots, int nb_cell, int nb_hist, const VcXb xb0,
                                                          const VcGate gate) {
  vc_spec_assume<SPEC>(d);
  // The peer-to-peer exchange folded into this launch (round 6; xb0.nslots > 0): every block passes the exchange's gate first --
  // block 0 raises this rank's flag (the kernel boundary behind phase A released its slot), waits for the peers' and publishes the
  // launch's verdict, the others wait for it -- and the readers below add the ranks' slots themselves (vc_xget): no launch for the sum
  VcXb xb = xb0;
  if (xb0.nslots > 0) xb.dead = vc_p2p_gate(gate.regions, gate.world, gate.rank, gate.step, gate.status, gate.timeout_ticks, gate.verdict, false);
  const long long s = *step_dev;
  if ((int)blockIdx.x < d.nb_post_gene) {
    VcOpt o;
    o.step_size = b.step_size[0];
    o.b1 = a.b1; o.b2 = a.b2; o.eps = a.eps; o.clip = a.clip; o.c2 = b.step_size[1]; o.wd = a.wd; o.frozen = a.frozen;
    vc_tail_gene_block<MQ, VC_PH_B>(d, b, P, G, a.m, a.v, a.header, blockIdx.x, s, seed, o, 0, xb);
    return;
  }
  const int oblk = blockIdx.x - d.nb_post_gene;
  if (d.hist_dense && oblk > nb_cell && oblk - nb_cell - 1 < nb_hist) {
    // dense histogram blocks: all 16 waves of the block share the count axis (a quarter of the chain of a 4-wave block)
    vc_omega_extra_block<false>(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_hist, oblk - nb_cell, VC_PH_B, xb, false, 1024);
    return;
  }
  if (threadIdx.x >= 256) return;          // K_omega's blocks are 256 threads wide: the other waves leave before any barrier
  vc_omega_block(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_cell, nb_hist, oblk, VC_PH_B, xb);
}

void vc_launch_phase_b(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                       const VcAdamArgs& a, double* loss_dev, long long loss_slots, int with_hist, const VcXb& xb,
                       hipStream_t st, const VcGate* gate_in) {
  const VcGate gate = gate_in ? *gate_in : VcGate{};
  const int nb_cell = d.model == VC_MODEL_VELOCITY ? (d.Nc + 255) / 256 : 0;
  const int nb_hist = with_hist ? vc_hist_blocks(d, b, 4) : 0;
  const int nb_eps = (int)((d.eps_total / 2 + 255) / 256);
  const dim3 grid(d.nb_post_gene + nb_cell + 1 + nb_hist + nb_eps), block(1024);
  // (the gene blocks of phase B neither reduce nor stage rows: the smallest row bound keeps their registers free)
  if (vc_spec_launch<VC_SPECK_SHARDED, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_phase_b_kernel<2, decltype(sp)::value>), grid, block, vc_hist_dyn_lds(d, with_hist, 1024), st, d, b, params, grad,
                           step_dev, seed, a, loss_dev, loss_slots, nb_cell, nb_hist, xb, gate);
      }))
    return;
  hipLaunchKernelGGL(vc_phase_b_kernel<2>, grid, block, vc_hist_dyn_lds(d, with_hist, 1024), st, d, b, params, grad, step_dev, seed, a,
                     loss_dev, loss_slots, nb_cell, nb_hist, xb, gate);
}

// ---------------------------------------------------------------------------------------------
// Phases A and B of a rank of a sharded run in ONE launch, the exchange at block granularity in between (round 6; opt-in with the
// peer-to-peer exchange: vc_tuning.p2p_one_launch).  The step becomes K_main -> this launch.  Block kinds in dispatch order:
//   gene block k        phase A of the block (partials into this rank's slot), publish flag k, wait for flag k of every rank, phase B of
//                       the block on the sum of the ranks' slices (vc_xget);
//   cell block c        256 cells: phase A (phi_xy update, next phase, basis, the block's row of d loglik / d nu_omega partials), publish
//                       flag NG + c, wait for ALL cell blocks of all ranks, K_omega's block of the same cells (nu_omega chain, omega_c);
//   the loss block      the rank's loss base, publish flag NG + NC (and the flags of cell blocks this rank does not have: the flag
//                       grid is rank-invariant), wait for every flag, the loss of the step;
//   histogram blocks    wait for the gene blocks, re-derive the shape_inv update from the snapshot those took, the next step's sums;
//   eps blocks          depend on nothing.
// A block that waits spins: every block it waits for must be resident or finished -- gene, cell and loss blocks come first in the
// grid and the host admits the launch only where they fit the chip at one 1024-thread block per CU (vc_svi_run_sharded).
// The same block code as phases A / B: the same bits as the three-launch sharded step (tests/test_hip_multiproc.py).
// ---------------------------------------------------------------------------------------------
template <int MQ, int SPEC = 0>
__global__ __launch_bounds__(1024) void vc_tail_x_kernel(const VcDims d, const VcBufs b, float* __restrict__ P, float* __restrict__ G,
                                                         const long long* __restrict__ step_dev, uint64_t seed, const VcAdamArgs a,
                                                         double* __restrict__ loss_dev, long long loss_slots, int nb_cell, int nc_cap,
                                                         int nb_hist, const VcXb xw, const VcXb xr0, const VcGateX gx) {
  vc_spec_assume<SPEC>(d);
  const long long s = *step_dev;
  const int NG = d.nb_post_gene;
  VcXb xr = xr0;
  VcOpt o;
  o.step_size = b.step_size[0];
  o.b1 = a.b1; o.b2 = a.b2; o.eps = a.eps; o.clip = a.clip; o.c2 = b.step_size[1]; o.wd = a.wd; o.frozen = a.frozen;
  const int bid = blockIdx.x;
  if (bid < NG) {
    vc_tail_gene_block<MQ, VC_PH_A>(d, b, P, G, a.m, a.v, a.header, bid, s, seed, o, 0, xw);
    vc_x_publish(gx, bid, 1, 1024);
    xr.dead = vc_x_wait(gx, bid, 1, 1024);
    vc_tail_gene_block<2, VC_PH_B>(d, b, P, G, a.m, a.v, a.header, bid, s, seed, o, 0, xr);
    return;
  }
  if (bid < NG + nb_cell) {
    if (threadIdx.x >= 256) return;        // 256 cells per block: the other waves leave before any barrier
    const int c = bid - NG;
    vc_tail_cell_block<VC_PH_A>(d, b, P, G, a.m, a.v, a.header, c, s, seed, o, 0, xw);
    vc_x_publish(gx, NG + c, 1, 256);
    if (d.model != VC_MODEL_VELOCITY) return;          // (the phase model has no nu_omega chain: nothing to wait for)
    xr.dead = vc_x_wait(gx, NG, nc_cap, 256);
    vc_omega_block(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_cell, nb_hist, c, VC_PH_B, xr);
    return;
  }
  const int xblk = bid - NG - nb_cell;     // 0 loss, 1 .. nb_hist histogram, then eps (as vc_omega_extra_block counts them)
  if (xblk == 0) {
    vc_tail_loss_base_block(d, b, s, xw);
    // this rank's loss flag + the flags of the cell blocks it does not have (a rank with fewer cells than the widest shard)
    vc_x_publish(gx, NG + nb_cell, nc_cap - nb_cell + 1, 1024);
    xr.dead = vc_x_wait(gx, 0, NG + nc_cap + 1, 1024);
    vc_omega_extra_block<false>(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_hist, 0, VC_PH_B, xr, false, 1024);
    return;
  }
  if (xblk - 1 < nb_hist) {
    xr.dead = vc_x_wait(gx, 0, NG, 1024);        // the gene blocks: their shape_inv snapshot and gradient partials
    if (d.hist_dense) {
      vc_omega_extra_block<false>(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_hist, xblk, VC_PH_B, xr, false, 1024);
      return;
    }
  }
  if (threadIdx.x >= 256) return;          // list-form histogram tasks and eps pairs: K_omega's 256-thread blocks
  vc_omega_extra_block<false>(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_hist, xblk, VC_PH_B, xr, false);
}

void vc_launch_tail_x(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                      const VcAdamArgs& a, double* loss_dev, long long loss_slots, int with_hist, const VcXb& xw, const VcXb& xr,
                      const VcGateX& gx, int nc_cap, hipStream_t st) {
  const int nb_cell = d.nb_tail_cell;                  // (tail_tc = 256: vc_svi_run_sharded checks) = K_omega's cell blocks
  const int nb_hist = with_hist ? vc_hist_blocks(d, b, 4) : 0;
  const int nb_eps = (int)((d.eps_total / 2 + 255) / 256);
  const dim3 grid(d.nb_post_gene + nb_cell + 1 + nb_hist + nb_eps), block(1024);
  const unsigned dyn = vc_hist_dyn_lds(d, with_hist, 1024);
  if (vc_spec_launch<VC_SPECK_SHARDED, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_tail_x_kernel<decltype(mq)::value, decltype(sp)::value>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a,
                           loss_dev, loss_slots, nb_cell, nc_cap, nb_hist, xw, xr, gx);
      }))
    return;
#define VC_TAILX_LAUNCH(MQ_) hipLaunchKernelGGL((vc_tail_x_kernel<MQ_>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_cell, nc_cap, nb_hist, xw, xr, gx)
  if (d.nq <= 2) VC_TAILX_LAUNCH(2);
  else if (d.nq <= 4) VC_TAILX_LAUNCH(4);
  else if (d.nq <= 6) VC_TAILX_LAUNCH(6);
  else VC_TAILX_LAUNCH(VC_MAXQ);
#undef VC_TAILX_LAUNCH
}

// ---------------------------------------------------------------------------------------------
// Tutorial flow on one rank (U-only kernel with pw_inline, phases / nu / shape_inv conditioned): K_tail's gene blocks and K_omega's
// blocks in ONE launch, side by side, with no dependency between them:
//   * the partials of d loglik / d nu_omega come from K_main (PWM), not from K_tail's cell blocks -- which, with the phases
//     conditioned, have nothing else to do and are not launched (their loss terms are constants both halves of LPF hold after
//     two ordinary steps; the nu_omega snapshot they took comes from K_omega's block 0 of the step before);
//   * LPP (the r-only likelihood terms the loss block adds) is step-invariant with shape_inv conditioned: the gene blocks
//     rewrite the same doubles while the loss block reads them;
//   * no histogram blocks (shape_inv conditioned), the eps blocks depend on nothing.
// vc_svi_run_fused runs the first two steps of every call the ordinary way and the rest through this launch.
// ---------------------------------------------------------------------------------------------
template <int MQ, int SPEC = 0>
__global__ __launch_bounds__(1024) void vc_tail_merged_kernel(const VcDims d, const VcBufs b, float* __restrict__ P,
                                                              float* __restrict__ G, const long long* __restrict__ step_dev,
                                                              uint64_t seed, const VcAdamArgs a, double* __restrict__ loss_dev,
                                                              long long loss_slots, int nb_ocell) {
  vc_spec_assume<SPEC>(d);
  const long long s = *step_dev;
  if ((int)blockIdx.x < d.nb_post_gene) {
    VcOpt o;
    o.step_size = b.step_size[0];
    o.b1 = a.b1; o.b2 = a.b2; o.eps = a.eps; o.clip = a.clip; o.c2 = b.step_size[1]; o.wd = a.wd; o.frozen = a.frozen;
    vc_tail_gene_block<MQ, VC_PH_ALL>(d, b, P, G, a.m, a.v, a.header, blockIdx.x, s, seed, o, 0, VcXb{});
    return;
  }
  if (threadIdx.x >= 256) return;          // K_omega's blocks are 256 threads wide: the other waves leave before any barrier
  vc_omega_block(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_ocell, 0, blockIdx.x - d.nb_post_gene, VC_PH_ALL, VcXb{});
}

void vc_launch_tail_merged(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                           const VcAdamArgs& a, double* loss_dev, long long loss_slots, hipStream_t st) {
  const int nb_ocell = (d.Nc + 255) / 256;
  const int nb_eps = (int)((d.eps_total / 2 + 255) / 256);
  const dim3 grid(d.nb_post_gene + nb_ocell + 1 + nb_eps), block(1024);
  if (vc_spec_launch<VC_SPECK_MERGED, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_tail_merged_kernel<decltype(mq)::value, decltype(sp)::value>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a,
                           loss_dev, loss_slots, nb_ocell);
      }))
    return;
  if (d.nq <= 2) hipLaunchKernelGGL((vc_tail_merged_kernel<2>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_ocell);
  else if (d.nq <= 4) hipLaunchKernelGGL((vc_tail_merged_kernel<4>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_ocell);
  else if (d.nq <= 6) hipLaunchKernelGGL((vc_tail_merged_kernel<6>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_ocell);
  else hipLaunchKernelGGL((vc_tail_merged_kernel<VC_MAXQ>), grid, block, 0, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_ocell);
}

// ---------------------------------------------------------------------------------------------
// The rest of a single-rank step in ONE launch (round 4): everything K_tail and K_omega do, side by side, with no dependency
// between the blocks of the launch:
//   * gene blocks: K_tail's, unchanged;
//   * cell blocks: K_tail's with the nu_omega chain INSIDE (vc_nuw_chain) -- the partials of d loglik / d nu_omega come out of
//     K_main (pw_inline: PWM), so the chain needs nothing from the other cell blocks; block 0 stores the nu_omega parameters
//     and the snapshot the next launch reads.  The phase model has no such chain;
//   * the loss block: every term of the finished step's loss was complete before the launch (LPF / LPR of its sample, its
//     histogram sums, K_main's partials);
//   * the histogram blocks: shape_inv(s) is being computed by the gene blocks of this very launch, so every task wave re-derives
//     that update from the snapshot of the launch before (vc_hist_rederive_block: the same bits) and writes the other half
//     of HL / HD;
//   * the eps blocks depend on nothing.
// Launch structures give the same bits: tests/test_hip_fused.py.
// ---------------------------------------------------------------------------------------------
template <int MQ, int SPEC = 0>
__global__ __launch_bounds__(1024) void vc_tail2_kernel(const VcDims d, const VcBufs b, float* __restrict__ P,
                                                        float* __restrict__ G, const long long* __restrict__ step_dev,
                                                        uint64_t seed, const VcAdamArgs a, double* __restrict__ loss_dev,
                                                        long long loss_slots, int nb_hist) {
  vc_spec_assume<SPEC>(d);
  const long long s = *step_dev;
  VcOpt o;
  o.step_size = b.step_size[0];
  o.b1 = a.b1; o.b2 = a.b2; o.eps = a.eps; o.clip = a.clip; o.c2 = b.step_size[1]; o.wd = a.wd; o.frozen = a.frozen;
  // Block order = dispatch order: the histogram blocks first (with the signatures compiled in they carry the longest chain of the
  // launch: a block whose counts go beyond 255 ends ~3 us behind the gene blocks), then gene blocks, cell blocks, the loss block, eps
  int bid = blockIdx.x, kind;      // kind: 0 gene, 1 cell, 2 loss / histogram / eps
  int xblk = 0;                    // index among the extras as vc_omega_extra_block counts them: 0 loss, 1 .. nb_hist histogram, then eps
#ifdef VC_HIST_LAST       // (A/B: the order of rounds 4 / early 5 -- histogram blocks behind the loss block)
  const int lead = 0;
#else
  const int lead = nb_hist;
#endif
  if (bid < lead) { kind = 2; xblk = 1 + bid; }
  else {
    bid -= lead;
    if (bid < d.nb_post_gene) kind = 0;
    else if (bid < d.nb_post_gene + d.nb_tail_cell) { kind = 1; bid -= d.nb_post_gene; }
    else { kind = 2; xblk = bid - d.nb_post_gene - d.nb_tail_cell; if (xblk > 0) xblk += lead; }
  }
#ifdef VC_DBG_SKIP      // measurement aid (profiles/r05_tail2_skip_variants.txt): which kind of block bounds the launch -- bit 0 gene, 1 cell, 2 the rest
  if ((VC_DBG_SKIP >> kind) & 1) return;
#endif
  if (kind == 0) vc_tail_gene_block<MQ, VC_PH_ALL>(d, b, P, G, a.m, a.v, a.header, bid, s, seed, o, 0, VcXb{});
  else if (kind == 1) vc_tail_cell_block<VC_PH_ALL, true>(d, b, P, G, a.m, a.v, a.header, bid, s, seed, o, 0, VcXb{});
  else vc_omega_extra_block<true>(d, b, P, G, s, seed, a, loss_dev, loss_slots, 0, nb_hist, xblk, VC_PH_ALL, VcXb{}, true, 1024);
}

void vc_launch_tail2(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                     const VcAdamArgs& a, double* loss_dev, long long loss_slots, int with_hist, hipStream_t st) {
  const int nb_hist = !with_hist ? 0 : (d.hist_dense ? (d.model == VC_MODEL_VELOCITY ? 2 : 1) * (d.Ng_pad / 64) + 4 * b.n_hc_split      // a block per (matrix, gene block) + the quarter blocks
                                                       : (b.n_tasks + 16 * VC_HIST_ROUNDS - 1) / (16 * VC_HIST_ROUNDS));
  const int nb_eps = (int)((d.eps_total / 2 + 1024 * VC_EPS_PER_THREAD - 1) / (1024 * VC_EPS_PER_THREAD));
  const dim3 grid(d.nb_post_gene + d.nb_tail_cell + 1 + nb_hist + nb_eps), block(1024);
  const unsigned dyn = vc_hist_dyn_lds(d, with_hist, 1024);
  if (vc_spec_launch<VC_SPECK_TAIL2, 0>(d.spec, [&](auto mq, auto sp) {
        hipLaunchKernelGGL((vc_tail2_kernel<decltype(mq)::value, decltype(sp)::value>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a,
                           loss_dev, loss_slots, nb_hist);
      }))
    return;
  if (d.nq <= 2) hipLaunchKernelGGL((vc_tail2_kernel<2>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_hist);
  else if (d.nq <= 4) hipLaunchKernelGGL((vc_tail2_kernel<4>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_hist);
  else if (d.nq <= 6) hipLaunchKernelGGL((vc_tail2_kernel<6>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_hist);
  else hipLaunchKernelGGL((vc_tail2_kernel<VC_MAXQ>), grid, block, dyn, st, d, b, params, grad, step_dev, seed, a, loss_dev, loss_slots, nb_hist);
}

void vc_launch_omega(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                     const VcAdamArgs& a, double* loss_dev, long long loss_slots, int boot, int with_hist, hipStream_t st) {
  const int nb_cell = d.model == VC_MODEL_VELOCITY ? (d.Nc + 255) / 256 : 0;
  const int nb_hist = with_hist ? vc_hist_blocks(d, b, 4) : 0;
  const int nb_eps = (int)((d.eps_total / 2 + 255) / 256);
  hipLaunchKernelGGL(vc_omega_kernel, dim3(nb_cell + 1 + nb_hist + nb_eps), dim3(256), vc_hist_dyn_lds(d, with_hist, 256), st, d, b,
                     params, grad, step_dev, seed, a, loss_dev, loss_slots, boot, nb_cell, nb_hist);
}
