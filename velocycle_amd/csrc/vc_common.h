// Shared definitions of the gfx950 engine: dimensions, buffer tables, wave-level helpers.
// Device code here is written for CDNA4 only (wave64, DPP row ops, SGPR-resident cell records).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/velocycle_hip.h"

#define VC_WAVES 4          // waves per workgroup of the likelihood kernel
// Arithmetic diet of the likelihood kernel's cell loop (round 3, profiles/r03_kmain.md section 5): each knob removes one packed
// operation per gene pair and cell.  They change what the cell record holds, so every kernel that writes a record sees them.
#ifndef VC_FOLD_LOG2E
#define VC_FOLD_LOG2E 1     // count noise models: the gene coefficients are scaled by log2 e once per gene and the record carries
#endif                      // cf * log2 e and omega * ln 2, so that eta * log2 e needs no multiply per (gene, cell)
#ifndef VC_OMEGA_CS
#define VC_OMEGA_CS 1       // velocity: the record carries k omega cos k phi, k omega sin k phi (no w * omega per gene pair)
#endif
#ifndef VC_HOIST_LB
#define VC_HOIST_LB 1       // S+U kernel, count noise: sum_c k_U log beta = log beta * (sum_c k_U) leaves the loop
#endif
#ifndef VC_NR_MERGE
#define VC_NR_MERGE 1       // negative-binomial U term: num * R formed once for a_U and w
#endif
#ifndef VC_PW_INLINE
#define VC_PW_INLINE 1      // tutorial flow on one rank (U-only kernel, phases conditioned): the per-workgroup partials of
#endif                      // d loglik / d nu_omega[j] = sum_c A3_c W_cj come out of the likelihood kernel itself -- the sum is linear
                            // in the gene blocks' shares of A3, and the kernel holds those per cell when it stores them (lane = cell,
                            // 64 cells per flush); W = D[x,c] zeta_omega_h(phi_c) of the wave's cells waits in the LDS
#define VC_PWQ 8            // coefficients the likelihood kernel carries (Nx * Nhw <= VC_PWQ, else K_tail's cell blocks do it): rows of
                            // 4 floats (pw_inline = 4) or 8 (pw_inline = 8: e.g. two samples x three harmonics of omega)
// (the staged W rows live in DYNAMIC shared memory: d.pw_slots float4 per wave, + 32 float4 of accumulators per wave for the
// S+U kernel; vc_main_dyn_lds() is what the launch asks for -- 0 bytes when pw_inline is off)
#ifndef VC_REC_PAD
#define VC_REC_PAD 2        // cell records are padded to a multiple of this many {x, x} pairs (2 = 16 bytes: the S+U kernel's record
                            // of 6 pairs at H = 1 then strides 48 bytes, not 64; measured 113.8 vs 114.1-116.7 us, profiles/r03_kmain.md)
#endif
#define VC_LOG2E 1.4426950408889634f
#define VC_LN2 0.6931471805599453f
#define VC_MAXH 3
#define VC_MAXNB 4
#define VC_MAX_NW 64        // max Nx*Nhw (angular-speed coefficients)
#define VC_MAX_RANK 8

#define VC_KIND_PHASE 0     // S likelihood only (phase model)
#define VC_KIND_VFULL 1     // S and U likelihoods, every gradient
#define VC_KIND_VU 2        // U likelihood only: phi, nu, dnu, shape_inv conditioned -> S term hoisted to setup

#define VC_LOG_2PI 1.8378770664093453f

struct VcDims {
  int Ng, Ng_pad, nGB;
  int gpl, gbw;           // genes per lane of the likelihood kernel (4 or 8), genes per gene block = 64*gpl
  int c16;                // 1: the blocked counts are uint16 (every count an integer <= 65535), 0: float32
  int Nc;                 // cells on this rank
  long long cell_offset;  // global index of the first local cell
  int H, Nh, Hw, Nhw, Nb, Nx, R, M, NW;   // M = Ng + Nx*Nhw, NW = Nx*Nhw
  int K;                  // Nh + Nb : expression-map coefficients per gene (harmonics, then batch offsets) = rows of the gene table in front of log beta
  int onehot;             // 1: the batch design matrix Db is one-hot (preprocessing.py:65-93 builds it so) and every workgroup of the
                          // likelihood kernel lies inside ONE batch: sum_b Db[b,c] dnu[b,g] = dnu[b(c),g] is added to the constant
                          // harmonic when the wave loads its latents -- the NB = 0 instantiation runs for ANY number of batches, at no
                          // cost per cell; d loglik / d dnu[b,g] = the sum of the constant harmonic's partial rows over the batch's
                          // workgroups (vc_dnu_range_sum)
  int Kq;                 // coefficient rows the likelihood kernel emits (K, or Nh when the batches are folded: onehot)
  int nbk;                // batch entries of a cell record (Nb, or 0: onehot / no batch offsets)
  int pw_inline;          // 4 | 8: K_main (U-only and S+U kernels) writes PWM rows of that many floats, K_omega / K_fin read them
                          // instead of the cell blocks' PW; 0: off
  int pw_slots;           // float4 slots per wave of K_main's staged W rows (>= cells per wave x float4 per row)
  int pw_lane;            // 1 (U-only kernel, one condition with D == 1, Hw <= H): W_c = (1, sin k phi_c, cos k phi_c) is the cell record itself --
                          // the partials are accumulated per lane, no W rows in the LDS, no per-cell rows stored
  int ctw;                // floats per cell record: {x,x} pairs of [sin k, cos k]*H, Db[Nb], omega, cf, S+U kernel: [k omega cos k,
                          // k omega sin k]*H (padded); omega and cf carry the scale factors of vc_rec_*_scale
  int model, guide, noise, with_dnu;
  unsigned cond;          // bit i set <=> site i conditioned
  int kind;               // VC_KIND_*
  int nq;                 // per-gene accumulator rows the likelihood kernel emits
  int nco;                // per-cell accumulator rows
  int n_chunks;           // cell chunks = workgroups per gene block
  int cw;                 // cells per wave (the widest pass)
  int pass_wgs;           // workgroups per dispatch pass of the likelihood kernel (= CUs): workgroup w runs in pass w / pass_wgs
  int pass_cw[4];         // cells per wave of the workgroups of pass 0..3 (later passes: as pass 3); all equal to cw unless the
                          // passes take unequal shares of the cells (vc_engine.hip, tiling)
  int hist_has_S, hist_has_U;
  int generic;            // 1: the run-time-sized kernel set (vc_generic_kernels.hip): a configuration outside the compiled fast set
  int hist_dense;         // 1: the histogram sums come from the dense tail-count tables (one task per gene and matrix, evaluated per
                          // gene block: vc_hist_dense_block); 0: from the (value, multiplicity) lists, one wave per task of <= 64 values
  int hist_par;           // 1: shape_inv is learned -- the fused steps keep the histogram sums of the sample of step s in half s & 1 of
                          // HL / HD (the launch that finishes step s - 1 reads half (s - 1) & 1 while its histogram blocks write half
                          // s & 1); 0: evaluated once (half 0)
  int nmat_r;             // matrices whose NB constant r*log r is evaluated per step
  float root_w;           // 1 on rank 0, 0 elsewhere: weight of replicated prior / entropy terms
  float gamma_alpha, gamma_beta, sigma_ln_s, sigma_ln_u, rho_mean, rho_std, rho_scale;
  long long poff[VC_P_COUNT];   // parameter offsets (floats) in the flat buffers, -1 if absent
  long long eoff[VC_E_COUNT];   // eps offsets
  long long eps_n_global, eps_total;
  int nb_pre_gene, nb_pre_cell, nb_post_gene, nb_post_cell, n_main_wg;
  int nb_tail_cell;       // fused pipeline: cell blocks of K_tail (tail_tc cells each)
  int tail_tc;            // cells per cell block of K_tail: 256 (one wave per SIMD, shortest chain) or 1024 (large shards)
  int nlpf;               // fused pipeline: loss slots per half of LPF = nb_post_gene + nb_tail_cell + 1
  float lgamma_alpha;     // lgamma(gamma_alpha) of the shape_inv prior, evaluated once on the host
  int spec;               // > 0: the small kernels of the fused steps run in the instantiation compiled for this configuration's
                          // SIGNATURE (vc_tail_spec.h: VC_SIGS[spec - 1]); 0: the run-time-flag kernels
};

// The SIGNATURE of a configuration: every field of VcDims the small kernels branch or loop on that does NOT depend on the size of
// the problem (cells, genes, ranks, tiling).  The kernels of the fused steps carry the code of every model, guide, noise model and
// conditioning pattern behind run-time flags -- 99 KB of instructions and 219 spilled SGPRs in the one-launch tail, most of which
// a given configuration never runs (round 5: profiles/r05_tail_spec.md).  For the signatures listed in vc_tail_spec.h the same
// source is compiled again with the signature's values as compile-time facts (__builtin_assume on exactly these fields, checked on
// the host by comparing the whole signature): dead branches go, loops unroll, the statements that remain are the same ones in the
// same order -- the same bits (tests/test_hip_fused.py holds the two against each other).
#define VC_SIG_FIELDS(X)                                                                                                             \
  X(model) X(guide) X(noise) X(with_dnu) X(kind) X(H) X(Nh) X(Hw) X(Nhw) X(Nb) X(Nx) X(R) X(NW) X(K) X(Kq) X(nbk) X(onehot)         \
  X(pw_inline) X(nq) X(nco) X(hist_has_S) X(hist_has_U) X(hist_dense) X(hist_par) X(nmat_r) X(generic)
struct VcSig {
#define VC_SIG_DECL(f) int f;
  VC_SIG_FIELDS(VC_SIG_DECL)
#undef VC_SIG_DECL
  unsigned cond;
};
#define VC_SIG_INTS 27        // ints of a signature as vc_dbg_signature lists them (the fields above in order, then cond)
static inline VcSig vc_sig_of(const VcDims& d) {
  VcSig s;
#define VC_SIG_GET(f) s.f = d.f;
  VC_SIG_FIELDS(VC_SIG_GET)
#undef VC_SIG_GET
  s.cond = d.cond;
  return s;
}

// dynamic shared memory of a K_main launch (bytes)
__host__ __device__ inline unsigned vc_main_dyn_lds(const VcDims& d) {
  if (!d.pw_inline) return 0u;
  return (unsigned)(VC_WAVES * (d.pw_slots + (d.kind == VC_KIND_VFULL ? 32 : 0)) * 16);
}

// Cell record: scale factors of its omega and cf entries, and its length in {x, x} pairs
__host__ __device__ inline float vc_rec_cf_scale(int noise) { return (VC_FOLD_LOG2E && noise != VC_NOISE_LOGNORMAL) ? VC_LOG2E : 1.f; }
__host__ __device__ inline float vc_rec_omega_scale(int noise) { return (VC_FOLD_LOG2E && noise != VC_NOISE_LOGNORMAL) ? VC_LN2 : 1.f; }
__host__ __device__ inline int vc_rec_pairs(int H, int Nb, bool full) { return 2 * H + Nb + 2 + ((full && VC_OMEGA_CS) ? 2 * H : 0); }

// Peer-mapped regions of the one-shot exchange (vc_p2p_exchange.hip): region[q] = rank q's region as mapped in THIS process
#define VC_P2P_MAX_RANKS 16
#define VC_P2P_FLAG_STRIDE 16      // 64-byte line per flag (in 4-byte words)
#define VC_P2P_POISON 0xFFFFFFFFu  // a rank that gave up on a step publishes this instead of a step number
struct VcP2p {
  void* region[VC_P2P_MAX_RANKS];
  int world = 0, rank = 0;
  int flag_words = 0;           // 4-byte words in front of the slots (world flags, one 64-byte line each)
  long long slot_floats = 0;    // floats per slot (= the exchange buffer, padded to 64 bytes)
};

// View of the exchange buffer of the sharded fused step (vc_svi_run_sharded; layout: include/velocycle_hip.h)
struct VcXb {
  float* x = nullptr;     // [0, header + n_global): gradient partials at the offsets of the gradient buffer;
  int pw_off = 0;         //   [pw_off, pw_off + pw_cap * NW): per-cell-block partials of d loglik / d nu_omega;
  int pw_cap = 0;         //   rows of that region (an upper bound of every rank's cell blocks: ceil(ceil(Nc / world) / 256))
  int loss_off = 0;       //   [loss_off, loss_off + 4 * (1 + nb_post_gene)): the rank's loss terms, four floats each
  float* sis = nullptr;   // [3][Ng_pad] snapshot {parameter, exp_avg, exp_avg_sq} of shape_inv taken by phase A (engine-owned)
  // Round 6, the peer-to-peer exchange FOLDED into phase B (no launch of its own): phase B's readers add the ranks' published
  // buffers themselves, in rank order (vc_xget) -- nslots = 0: `x` holds the summed buffer (RCCL / torch / the separate exchange kernel)
  int nslots = 0;
  int xmode = 0;          // 1: phases A and B run in ONE launch (vc_tail_x_kernel): what a block of the launch wrote and ANOTHER block reads
                          // (the shape_inv snapshot `sis`) is read with cache-bypassing loads too -- no kernel boundary lies in between
  int dead = 0;           // the launch's verdict (vc_p2p_gate): a peer never published -> every read is NaN (the step is poisoned)
  const float* const* slots = nullptr;    // DEVICE table [nslots]: rank q's published buffer of this step's parity, as mapped in this process
                                           // (a table in memory, not an array in the kernel arguments: indexing those by a run-time rank
                                           // puts the whole argument struct into scratch -- phase B ran 2x slower with it)
};
// what a launch needs to run the exchange's publish / wait protocol itself (vc_p2p_gate)
struct VcGate {
  void* const* regions = nullptr;      // DEVICE table [world]: rank q's region as mapped in this process
  int world = 0, rank = 0;
  long long step = 0;
  long long* status = nullptr;
  unsigned long long timeout_ticks = 0;
  unsigned long long* verdict = nullptr;
};
#ifdef __HIPCC__
// one float of a peer's published buffer: a system-scope load that bypasses the caches (the peer rewrote the slot two steps ago);
// compiler-visible, so that the N loads of a sum are in flight together
__device__ __forceinline__ float vc_xload(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
// element i of the exchange buffer summed over the ranks: read from the summed buffer, or added up here in rank order (the order of
// the exchange kernel: identical bits on every rank)
// (the table of slot pointers is read through the SCALAR cache -- it is written once, by the host, when the regions are connected: as
// plain loads hipcc put `s_waitcnt vmcnt(0)` between every pointer and its element, which also waited for the previous rank's element:
// 2 N dependent round trips per element instead of one; found in the ISA of phase B, round 6)
__device__ __forceinline__ const float* vc_xslot(const void* table, int q) {
  typedef const __attribute__((address_space(4))) unsigned long long* ctab;
  return reinterpret_cast<const float*>(((ctab)table)[q]);
}
__device__ __forceinline__ float vc_xget(const VcXb& xb, long long i) {
  if (xb.nslots == 0) return xb.x[i];
  if (xb.dead) return __builtin_nanf("");
  float acc = 0.f;
  for (int q0 = 0; q0 < xb.nslots; q0 += 4) {      // four ranks' loads in flight per trip
    const float* sl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) sl[u] = vc_xslot(xb.slots, q0 + u < xb.nslots ? q0 + u : q0);
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = vc_xload(sl[u] + i);          // (a rank beyond the last: rank q0's element again, dropped)
#pragma unroll
    for (int u = 0; u < 4; ++u) if (q0 + u < xb.nslots) acc += v[u];
  }
  return acc;
}

// (xmode) a float another block of the SAME launch wrote and released (vc_x_publish): bypass the caches; else a plain load
__device__ __forceinline__ float vc_xsis(const VcXb& xb, size_t i) { return xb.xmode ? vc_xload(xb.sis + i) : xb.sis[i]; }
// (xmode) what phase A hands to the exchange -- slot elements, the shape_inv snapshot -- goes out as write-through system-scope stores:
// when the wave's vmcnt has drained they have reached memory, and the block's flag may be raised WITHOUT a release fence (a release at
// system scope writes the XCD's whole L2 back: measured + 12 us per step with one fence per block); else plain stores (the kernel
// boundary behind phase A releases them)
__device__ __forceinline__ void vc_xstore(float* p, float v, int xmode) {
  if (xmode) __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else *p = v;
}
__device__ __forceinline__ void vc_xput(const VcXb& xb, long long i, float v) { vc_xstore(xb.x + i, v, xb.xmode); }

// ---- the exchange at BLOCK granularity (round 6, vc_tail_x_kernel: phases A and B of a rank in ONE launch) -------------------------
// Every block k of the launch (gene blocks, cell blocks, the loss block) owns ONE flag per rank: region[q].bflags[r][k] = what rank r's
// block k has published, as seen in rank q's region.  A block writes its partials into its rank's slot (phase A), releases them,
// raises its flag in EVERY region and waits only for the flags it depends on: a gene block for the same gene block of every rank, a
// cell block for all cell blocks of all ranks (the nu_omega gradient), the loss block for everything.  No grid-wide barrier.
struct VcGateX {
  void* const* regions = nullptr;   // DEVICE table [world]
  int world = 0, rank = 0;
  long long step = 0;
  long long* status = nullptr;
  unsigned long long timeout_ticks = 0;
  long long bflag_off = 0;          // 4-byte words from a region's base to bflags[world][nblk]
  int nblk = 0;                     // flags per rank
};
// Release what this block wrote (all of its live threads call; nthr of them) and raise flags [k0, k0 + nk) of this rank in every region.
__device__ __forceinline__ void vc_x_publish(const VcGateX& g, int k0, int nk, int nthr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // every store of the block has left its wave
  const unsigned want = (g.status && g.status[2] != 0) ? VC_P2P_POISON : (unsigned)(g.step + 1);
  if ((int)threadIdx.x < 64) {
    // (no fence: everything the peers / the other blocks read was stored write-through -- vc_xput -- and has drained)
    for (int i = threadIdx.x; i < g.world * nk; i += 64) {
      const int q = i / nk, j = i % nk;
      unsigned* f = reinterpret_cast<unsigned*>(g.regions[q]) + g.bflag_off + (size_t)g.rank * g.nblk + k0 + j;
      __hip_atomic_store(f, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
// Wait (bounded) until flags [k0, k0 + nk) of EVERY rank have reached this step in this rank's own region; 1 = dead (time-out or a
// poisoned peer: sticky in status[2]), block-uniform.  All nthr live threads of the block call.
__device__ __forceinline__ int vc_x_wait(const VcGateX& g, int k0, int nk, int nthr) {
  __shared__ int sm_xdead;
  if (threadIdx.x == 0) sm_xdead = (g.status && g.status[2] != 0) ? 1 : 0;
  __syncthreads();
  const unsigned want = (unsigned)(g.step + 1);
  const unsigned* mine = reinterpret_cast<const unsigned*>(g.regions[g.rank]) + g.bflag_off;
  for (int i = threadIdx.x; i < g.world * nk; i += nthr) {
    const int q = i / nk, j = i % nk;
    const unsigned* f = mine + (size_t)q * g.nblk + k0 + j;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v == VC_P2P_POISON) { sm_xdead = 1; break; }
      if ((int)(v - want) >= 0) break;
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > g.timeout_ticks) { sm_xdead = 1; break; }
    }
  }
  __syncthreads();
  const int dead = sm_xdead;
  if (dead && threadIdx.x == 0 && g.status && g.status[2] == 0) g.status[2] = g.step + 1;
  return dead;
}

// The publish / wait protocol of the one-shot exchange (vc_p2p_exchange.hip has the story), for every block of a launch whose
// predecessor on the stream wrote this rank's slot: block 0 raises this rank's flag of `step` in every region (the kernel boundary
// in front of the launch released the slot's plain stores), waits -- bounded -- for every peer's flag in its OWN region and publishes
// the launch's ONE verdict; the other blocks wait for that verdict.  Returns 1 when the step is dead (a peer never published or had
// given up: sticky in status[2], poison raised in every region), else 0; block-uniform.  Ends with a system-scope acquire.
__device__ __forceinline__ int vc_p2p_gate(void* const* __restrict__ regions, int world, int rank, long long step,
                                           long long* __restrict__ status, unsigned long long timeout_ticks,
                                           unsigned long long* __restrict__ verdict, const bool acquire = true) {
  // acquire = false (phase B with the exchange folded in): the blocks that only wait for the verdict poll it with relaxed
  // cache-bypassing loads and the gate ends WITHOUT the system-scope acquire -- an acquire at agent / system scope invalidates the
  // XCD's whole L2, and ~1000 waves of a 60-block launch doing that while other blocks stream K_main's partials made the launch
  // 150 us slower (measured); every read of a peer's slot behind this gate bypasses the caches by itself (vc_xload)
  struct { void* const* region; int world, rank; } p = {regions, world, rank};
  const unsigned want = (unsigned)(step + 1);
  __shared__ int sm_dead;
  if (threadIdx.x == 0) sm_dead = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    // sticky: once this rank has poisoned a step it never sums again (the caller sees VC_ERR_STATE at its next status check)
    const bool already = status && status[2] != 0;
    if (threadIdx.x < (unsigned)p.world) {
      __atomic_thread_fence(__ATOMIC_RELEASE);       // system scope
      unsigned* f = reinterpret_cast<unsigned*>(p.region[threadIdx.x]) + (size_t)p.rank * VC_P2P_FLAG_STRIDE;
      __hip_atomic_store(f, already ? VC_P2P_POISON : want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (already && threadIdx.x == 0) sm_dead = 1;
    if (threadIdx.x < (unsigned)p.world && !already) {
      const unsigned* f = reinterpret_cast<const unsigned*>(p.region[p.rank]) + (size_t)threadIdx.x * VC_P2P_FLAG_STRIDE;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == VC_P2P_POISON) { sm_dead = 1; break; }              // the peer gave up on a step: so does this rank
        if ((int)(v - want) >= 0) break;
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { sm_dead = 1; break; }
      }
    }
    __syncthreads();
    if (sm_dead && !already) {
      if (threadIdx.x == 0 && status && status[2] == 0) status[2] = step + 1;          // a peer never arrived / had given up
      if (threadIdx.x < (unsigned)p.world) {
        unsigned* f = reinterpret_cast<unsigned*>(p.region[threadIdx.x]) + (size_t)p.rank * VC_P2P_FLAG_STRIDE;
        __hip_atomic_store(f, VC_P2P_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (threadIdx.x == 0) {
      const unsigned long long vd = ((unsigned long long)want << 1) | (unsigned long long)(sm_dead ? 1 : 0);
      if (acquire) __hip_atomic_store(verdict, vd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else __hip_atomic_store(verdict, vd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (write-through: the pollers bypass the caches)
    }
  } else {
    // the launch's one verdict (block 0 is dispatched first and waits for nothing inside this launch)
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned long long v = acquire ? __hip_atomic_load(verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                             : __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(v >> 1) == want) { sm_dead = (int)(v & 1ull); break; }
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2 * timeout_ticks + 100000000ull) { sm_dead = 1; break; }
      }
    }
    __syncthreads();
  }
  if (acquire) __atomic_thread_fence(__ATOMIC_ACQUIRE);         // system scope: nothing of the peers' slots may come from a stale line
  return sm_dead;
}
#endif

struct VcBufs {
  // immutable inputs
  const float *S, *U;                       // blocked counts [nGB][Nc][gbw], float32 or (d.c16) uint16
  const float *cf, *Dm, *Dbm, *pxy;         // (Nc), (Nx,Nc), (Nb,Nc), (Nc,2)
  const float *mu_nu, *sd_nu, *mu_g, *sd_g, *mu_b, *sd_b, *mu_w, *sd_w, *sd_dnu;
  const float *cnd[VC_SITE_COUNT];          // conditioned values per site (or nullptr)
  const int *h_ptr;                         // histogram CSR: [2*Ng+1], S genes then U genes
  const int *h_task;                        // histogram tasks (<= 64 entries each): {gene, matrix, begin, end} x n_tasks
  const int *h_tptr;                        // [Ng+1] first task of every gene (tasks are sorted by gene)
  int n_tasks;
  const float *h_val, *h_cnt;
  // dense histograms (d.hist_dense; vc_host_logic.h: vc_build_dense_hist): tail counts C_j of every gene, [gene block of 64][j][gene]
  const float* HC;                          // rows of 64 floats
  const int* hc_off;                        // [2][Ng_pad / 64] first row of a gene block (matrix S, then U)
  const int* hc_rows;                       // [2][Ng_pad / 64] rows of a gene block = its largest count (row j: cells with count > j)
  const int* hc_rows_q;                     // [2][Ng_pad / 64][4] the same per quarter (16 genes) of a gene block
  const int* hc_split;                      // [n_hc_split] gene blocks (matrix * Ng_pad / 64 + block) with hc_rows > VC_HIST_SPLIT_ROWS: the one-launch
  int n_hc_split;                           // tail evaluates them in four quarter blocks each (vc_hist_dense16_finish<true>)
  const int *wg_tile;                       // [n_main_wg][4] {first cell of wave 0, cells per wave, batch, end of the workgroup's cells} of the
                                            // likelihood kernel's workgroups (cells = POSITIONS of the blocked layout: cell_pos)
  const int *bat_chunk;                     // onehot: [nGB][Nb + 1] first chunk of batch q among the chunks of a gene block (chunks of a batch are
                                            // consecutive); entry Nb = n_chunks
  const int *cell_pos;                      // position of cell c in the blocked counts / cell table / per-cell partial rows (cells ordered by
                                            // batch when the batches are not contiguous); nullptr: the identity
  const float *gene_sum_u;                  // [Ng_pad] sum over this rank's cells of the unspliced counts of every gene (count noise)
  // per-step workspaces
  float *eps_used;
  float *GT;                                // gene table [K+3][Ng_pad]
  float *CT;                                // cell table [Nc][ctw]
  float *lat[VC_SITE_COUNT];                // site values of the last step
  float *lat_delta, *lat_sgam;              // LRMN: (M) deviation W eps_W + sqrt(D) eps_D, (Ng) marginal std of log gamma
  float *lat_phi, *lat_omega, *lat_domega;  // (Nc)
  float *GO;                                // per-gene partial sums [n_chunks][nq][Ng_pad]
  float *CO;                                // per-cell partial sums [nGB][nco][Nc]
  float *LO;                                // likelihood partial per main workgroup
  double *LP;                               // loss partials of pre (nb_pre_gene + nb_pre_cell) and post_gene blocks
  float *PW;                                // [nb_post_cell][NW] partial angular-speed gradients
  float *PWM;                               // [n_main_wg][pw_inline] the same partials, per workgroup of K_main
  float *WT;                                // [Nc][pw_inline] W_cj = D[x,c] zeta_omega_h(phi_c), j = x * Nhw + h (j >= NW: 0)
  double *HL, *HD;                          // [2][n_tasks] per histogram task: sum cnt*(lgamma(r+k)-lgamma(r)), sum cnt*(psi(r+k)-psi(r));
                                            // the unfused kernels use half 0, the fused steps the half of the sample's step (d.hist_par)
  double const_loss;                        // step-invariant part of the loss
  long long* status;                        // [0] number of steps with a non-finite loss, [1] 1 + index of the first one
  // fused single-rank pipeline (vc_svi_step_fused)
  long long* step_ctr;                      // non-null: K_main advances this device step counter (block 0) ...
  float* step_size;                         // ... and leaves the optimiser's step size [0] and second-moment bias correction [1] of the new step here (fp64 math, once)
  double adam_lr0, adam_lrd_l, adam_b1l, adam_b2l;
  int adam_kind;                            // VC_OPT_*: which step size / bias correction K_main leaves in step_size[0..1]
  double* LPF;                              // [2][nlpf] prior / guide loss terms of the samples of step s in half s & 1
  double* LPP;                              // [nb_post_gene] r-only likelihood terms of the gene blocks (phase A of the sharded step)
  double* LPR;                              // [2][nb_post_gene] -nmat_r Nc sum_g r log r of the sample of step s in half s & 1 (written when
                                            // the sample is drawn: the loss block of the launch that finishes the step reads it)
  float* SIS;                               // [2][4][Ng_pad] shape_inv {parameter, exp_avg, exp_avg_sq, value} as they stand after the
                                            // launch that drew the sample of step s (half s & 1): the histogram blocks of the NEXT
                                            // launch re-derive the update from it while the gene blocks of that launch rewrite the originals
  float* NWS;                               // [4][VC_MAX_NW * (VC_MAX_RANK + 2)] snapshot of the nu_omega parameters, moments, value
  float* EPS;                               // [3][eps_total] ring of standard-normal draws: slot (step % 3) holds the draws of `step`
#ifdef VC_DBG_TIMES
  unsigned long long* dbg;                  // measurement aid: 4 wall-clock stamps per wave of K_main
#endif
};

#ifdef __HIPCC__
#ifdef VC_DBG_TIMES      // measurement aid: per-block wall-clock stamps of the small kernels (kid 0 pre, 1 post, 2 fin)
#define VC_KSTAMP(kid, k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) b.dbg[(size_t)d.n_main_wg * 32 + ((size_t)(kid) * 4096 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
// per-WAVE stamps of the fused kernels (kid 0 K_tail, 1 K_omega), 8 slots per wave, behind the regions above
#define VC_WSTAMP(kid, k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) b.dbg[(size_t)d.n_main_wg * 32 + 3 * 4096 * 8 + (((size_t)(kid) * 4096 + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (k)] = wall_clock64(); } while (0)
#define VC_DBG_WORDS(nwg) ((size_t)(nwg) * 32 + 3 * 4096 * 8 + 2 * 4096 * 16 * 8)
#else
#define VC_KSTAMP(kid, k) do {} while (0)
#define VC_WSTAMP(kid, k) do {} while (0)
#endif
// The omega entries of a cell record (every value twice {x, x}: an SGPR pair is a packed operand of the likelihood kernel);
// sk / ck = sin, cos of k phi, k = 1..H
__device__ __forceinline__ void vc_rec_put_omega(float2* ct, const VcDims& d, float omega, const float* sk, const float* ck) {
  const int nbk = d.nbk;
  const float oz = omega * vc_rec_omega_scale(d.noise);
  ct[2 * d.H + nbk] = make_float2(oz, oz);
  if (VC_OMEGA_CS && d.kind == VC_KIND_VFULL) {          // read by the S+U kernel only; the other kinds keep the short record
#pragma unroll
    for (int k = 0; k < VC_MAXH; ++k)       // compile-time indices: sk / ck stay in registers
      if (k < d.H) {
        const float w = (float)(k + 1) * omega;
        ct[2 * d.H + nbk + 2 + 2 * k] = make_float2(w * ck[k], w * ck[k]);
        ct[2 * d.H + nbk + 3 + 2 * k] = make_float2(w * sk[k], w * sk[k]);
      }
  }
}
// position of cell c in the likelihood kernel's layout (blocked counts, cell table, per-cell partial rows, W table)
// sin phi, cos phi of the packed direction phi = atan2(y, x) (utils.py:488-506) WITHOUT the round trip through the angle: the
// reference rounds phi to float32 (up to 2.4e-7 absolute near +-pi) and then takes cos / sin of that; y / r and x / r are within
// ~1.5 ulp of the exact values.  It matters where ElogU's relu kink amplifies: d loglik / d z carries 1 / (z + 1e-5) with
// z = nu . zeta'(phi) omega + gamma, so an error of 1e-7 in zeta' is a per-cent error of that element's term (round 6:
// profiles/r06_kink_error.md).  Division and square root are correctly rounded (hipcc's default).  atan2(0, 0) = 0.
__device__ __forceinline__ void vc_dir_sincos(float x, float y, float* s1, float* c1) {
#ifdef VC_DIR_VIA_ANGLE       // (A/B build of profiles/tools/kink_error.py: rounds 1-5, the reference's own order of operations)
  sincosf(atan2f(y, x), s1, c1);
  return;
#endif
  const float r = sqrtf(fmaf(x, x, y * y));
  if (r == 0.f) { *s1 = 0.f; *c1 = 1.f; return; }
  *s1 = y / r;
  *c1 = x / r;
}
__device__ __forceinline__ int vc_pos(const VcBufs& b, int c) { return b.cell_pos ? b.cell_pos[c] : c; }
// Row of cell c (at position cp) of the W table of the U-only / S+U kernels (pw_inline): sk / ck = sin, cos of k phi_c up to Hw
__device__ __forceinline__ void vc_put_w(const VcDims& d, const VcBufs& b, int c, int cp, const float* sk, const float* ck) {
  if (!(d.pw_inline && (d.kind == VC_KIND_VU || d.kind == VC_KIND_VFULL))) return;
  for (int xq = 0; xq < d.Nx; ++xq) {
    const float dx = b.Dm[(size_t)xq * d.Nc + c];
#pragma unroll
    for (int h = 0; h < 2 * VC_MAXH + 1; ++h)
      if (h < d.Nhw) {
        const float z = (h == 0) ? 1.f : ((h & 1) ? sk[(h - 1) >> 1] : ck[(h - 1) >> 1]);
        b.WT[(size_t)cp * d.pw_inline + xq * d.Nhw + h] = dx * z;
      }
  }
}
// The second-stage reduction of K_main's gene-level partial rows by the 16 waves of a gene block (K_post, the tails' gene blocks, phase A):
// wave w adds the rows of chunks first, first + stride, ... < end in that order; the block's sum is then sm[0] + ... + sm[15].
// Plain: first = w, stride = 16 over all chunks.  One-hot batches, 2 <= Nb <= VC_WALK_MAXNB (round 6): the waves are dealt out to
// the BATCHES (16 / Nb each, the last batch takes the rest) and walk only their batch's chunk range -- the sum of the constant
// harmonic's row over batch q, i.e. d loglik / d dnu[q, g], is then the sum of that batch's waves' partials in the LDS: no second pass
// over the rows (round 5 summed the range again behind the barrier: two dependent round trips of 32 rows on one wave per batch).
// ONE association for every caller, like vc_dnu_range_sum.
#define VC_WALK_WAVES 16
#define VC_WALK_MAXNB 8
struct VcChunkWalk { int first, stride, end; };
__device__ __forceinline__ bool vc_walk_by_batch(const VcDims& d) { return d.onehot && d.Nb >= 2 && d.Nb <= VC_WALK_MAXNB; }
__device__ __forceinline__ void vc_walk_waves_of(const VcDims& d, int q, int* w0, int* nw) {      // the waves of batch q
  const int wb = VC_WALK_WAVES / d.Nb;
  *w0 = q * wb;
  *nw = (q == d.Nb - 1) ? VC_WALK_WAVES - q * wb : wb;
}
// the same walk where the lanes of a wave hold genes of DIFFERENT gene blocks of the likelihood kernel (the histogram blocks that
// re-derive the shape_inv update lane by lane): ordinary loads of the batch's range
__device__ __forceinline__ VcChunkWalk vc_chunk_walk_lane(const VcDims& d, const VcBufs& b, int g, int wave) {
  VcChunkWalk k;
  if (vc_walk_by_batch(d)) {
    const int wb = VC_WALK_WAVES / d.Nb;
    int q = wave / wb;
    if (q > d.Nb - 1) q = d.Nb - 1;
    int w0, nw;
    vc_walk_waves_of(d, q, &w0, &nw);
    const int* bc = b.bat_chunk + (size_t)(g / d.gbw) * (d.Nb + 1) + q;
    k.first = bc[0] + (wave - w0);
    k.stride = nw;
    k.end = bc[1];
  } else {
    k.first = wave; k.stride = VC_WALK_WAVES; k.end = d.n_chunks;
  }
  return k;
}
__device__ __forceinline__ VcChunkWalk vc_chunk_walk(const VcDims& d, const VcBufs& b, int g, int wave) {
  VcChunkWalk k;
  if (vc_walk_by_batch(d)) {
    const int wb = VC_WALK_WAVES / d.Nb;
    int q = wave / wb;
    if (q > d.Nb - 1) q = d.Nb - 1;
    int w0, nw;
    vc_walk_waves_of(d, q, &w0, &nw);
    // (wave-uniform: the 64 genes of the block lie in one gene block of the likelihood kernel; a scalar load like the tile table's)
    typedef const __attribute__((address_space(4))) int* ciptr;
    ciptr bc = (ciptr)(const void*)(b.bat_chunk + (size_t)__builtin_amdgcn_readfirstlane(g / d.gbw) * (d.Nb + 1) + q);
    k.first = bc[0] + (wave - w0);
    k.stride = nw;
    k.end = bc[1];
  } else {
    k.first = wave; k.stride = VC_WALK_WAVES; k.end = d.n_chunks;
  }
  return k;
}
// onehot: d loglik / d dnu[q, g] = the sum of the constant harmonic's partial row (GO row 0) over the workgroups of batch q of the
// likelihood kernel's gene block that holds gene g -- chunks [c0, c1) of that gene block, added in chunk order (one fixed
// association for every caller: K_post, K_tail, phase A), 32 chunks requested per trip
// The same sum in two halves (the one-launch tail's gene blocks): `issue` only REQUESTS the first 32 chunk rows of the range (the
// range itself was fetched at the top of the block), `finish` adds them in chunk order and takes further trips for a longer range --
// the same association as vc_dnu_range_sum.
#define VC_DNU_PRE 32
struct VcDnuPre { float v[VC_DNU_PRE]; int c0, c1; };
__device__ __forceinline__ void vc_dnu_range_issue(const VcDims& d, const VcBufs& b, int g, VcDnuPre& h) {
  const float* __restrict__ go = b.GO + g;
  const size_t stride = (size_t)d.nq * d.Ng_pad;
#pragma unroll
  for (int u = 0; u < VC_DNU_PRE; ++u)        // (an EMPTY range -- a batch without cells on this rank -- may start at n_chunks: row 0 instead)
    h.v[u] = go[(size_t)(h.c0 + u < h.c1 ? h.c0 + u : (h.c0 < h.c1 ? h.c0 : 0)) * stride];
}
__device__ __forceinline__ float vc_dnu_range_finish(const VcDims& d, const VcBufs& b, int g, const VcDnuPre& h) {
  const float* __restrict__ go = b.GO + g;
  const size_t stride = (size_t)d.nq * d.Ng_pad;
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < VC_DNU_PRE; ++u) if (h.c0 + u < h.c1) acc += h.v[u];
  for (int ch0 = h.c0 + VC_DNU_PRE; ch0 < h.c1; ch0 += VC_DNU_PRE) {
    float v[VC_DNU_PRE];
#pragma unroll
    for (int u = 0; u < VC_DNU_PRE; ++u) v[u] = go[(size_t)(ch0 + u < h.c1 ? ch0 + u : ch0) * stride];
#pragma unroll
    for (int u = 0; u < VC_DNU_PRE; ++u) if (ch0 + u < h.c1) acc += v[u];
  }
  return acc;
}
__device__ __forceinline__ float vc_dnu_range_sum(const VcDims& d, const VcBufs& b, int g, int q) {
  const int gbm = g / d.gbw;
  const int c0 = b.bat_chunk[gbm * (d.Nb + 1) + q], c1 = b.bat_chunk[gbm * (d.Nb + 1) + q + 1];
  const float* __restrict__ go = b.GO + g;
  const size_t stride = (size_t)d.nq * d.Ng_pad;
  float acc = 0.f;
  constexpr int UB = 32;
  for (int ch0 = c0; ch0 < c1; ch0 += UB) {
    float v[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) v[u] = go[(size_t)(ch0 + u < c1 ? ch0 + u : ch0) * stride];
#pragma unroll
    for (int u = 0; u < UB; ++u) if (ch0 + u < c1) acc += v[u];
  }
  return acc;
}
// ---------------------------------------------------------------------------------------------
// wave64 reductions with DPP row operations; the total lands in lane 63.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float vc_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, true));
}

__device__ __forceinline__ float vc_wave_sum_lane63(float v) {
  v += vc_dpp<0x111, 0xf>(v);   // row_shr:1
  v += vc_dpp<0x112, 0xf>(v);   // row_shr:2
  v += vc_dpp<0x114, 0xf>(v);   // row_shr:4
  v += vc_dpp<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds its row sum
  // row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 as fused adds that leave the other rows
  // untouched (row_mask), instead of v_mov_dpp into a zeroed temporary + v_add.  hipcc inserts no wait
  // states inside an asm statement: a VALU write needs 2 of them before a DPP read (s_nop 1).
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc"
               : "+v"(v));
  return v;
}

__device__ __forceinline__ float vc_wave_sum(float v) {   // broadcast as a wave-uniform scalar
  v = vc_wave_sum_lane63(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ double vc_wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// the same sum with DPP row operations on the two halves of the double (3 VALU per step instead of two LDS-crossbar
// permutes with their latency): the total lands in LANE 63 only
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double vc_dpp_d(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double vc_wave_sum_d63(double v) {
  v += vc_dpp_d<0x111, 0xf>(v);   // row_shr:1
  v += vc_dpp_d<0x112, 0xf>(v);   // row_shr:2
  v += vc_dpp_d<0x114, 0xf>(v);   // row_shr:4
  v += vc_dpp_d<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds its row sum
  v += vc_dpp_d<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3 (other rows add 0)
  v += vc_dpp_d<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
  return v;
}

// block sum in double for <=1024 threads; result valid in thread 0
__device__ __forceinline__ double vc_block_sum_d(double v, double* sm /* [16] */) {
  v = vc_wave_sum_d(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) t += sm[i];
  __syncthreads();
  return t;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG + Box-Muller: eps(index) for (seed, step), identical on every rank.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void vc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                          uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float vc_philox_normal(uint64_t seed, long long step, long long idx) {
  uint32_t o[4];
  const uint64_t blk = (uint64_t)idx >> 1;
  vc_philox((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)step, (uint32_t)((uint64_t)step >> 32),
            (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
  const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  // hardware log2 / sin / cos (v_log_f32, v_sin_f32, v_cos_f32 take revolutions): plenty for a noise draw
  const float rad = sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(u1));
  return rad * ((idx & 1) ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
}

// both normals of one Philox block (the even / odd index of a pair: cos and sin branch of the same Box-Muller draw):
// n0 == vc_philox_normal(seed, step, 2 * blk), n1 == vc_philox_normal(seed, step, 2 * blk + 1), at the cost of one
__device__ __forceinline__ void vc_philox_normal2(uint64_t seed, long long step, uint64_t blk, float& n0, float& n1) {
  uint32_t o[4];
  vc_philox((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)step, (uint32_t)((uint64_t)step >> 32),
            (uint32_t)seed, (uint32_t)(seed >> 32), o);
  const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float rad = sqrtf(-2.0f * 0.6931471805599453f * __builtin_amdgcn_logf(u1));
  n0 = rad * __builtin_amdgcn_cosf(u2);
  n1 = rad * __builtin_amdgcn_sinf(u2);
}

// eps value `li` (local index in this rank's eps vector, global index gi for the counter RNG)
__device__ __forceinline__ float vc_eps(const float* __restrict__ eps_in, float* __restrict__ eps_used,
                                        uint64_t seed, long long step, long long li, long long gi) {
  const float e = eps_in ? eps_in[li] : vc_philox_normal(seed, step, gi);
  eps_used[li] = e;
  return e;
}

__device__ __forceinline__ float vc_normal_lp(float x, float mu, float sd) {
  const float z = (x - mu) / sd;
  return -0.5f * z * z - logf(sd) - 0.5f * VC_LOG_2PI;
}

// lgamma(x+k) - lgamma(x) and digamma(x+k) - digamma(x) for x > 0, k >= 0, without ever forming the two
// large values: shift x up to y >= 8 with the recurrence (as logs of ratios / a sum of reciprocal
// differences), then the Stirling series written as a difference:
//   lgamma(y+k) - lgamma(y) = (y - 1/2) log1p(k/y) + k log(y+k) - k + S(y+k) - S(y)
//   psi(y+k)    - psi(y)    = log1p(k/y) - (1/(y+k) - 1/y)/2 - (T(y+k) - T(y))
// Evaluated in fp32 (no cancellation is left in this form: relative error ~1e-6, checked against
// scipy in tests/test_oracle_golden.py through the Python twin of this routine); the caller
// accumulates cnt * value over the histogram in fp64.
__device__ __forceinline__ void vc_lgamma_digamma_diff(float x, float k, float& dl, float& dd) {
  float lp = 0.f, rs = 0.f, y = x;
  if (x < 8.f) {
    const int n = (int)ceilf(8.f - x);
    for (int j = 0; j < n; ++j) {
      const float a = x + (float)j, bb = a + k;
      const float ib = 1.0f / bb;
      lp += logf(a * ib);              // log((x+j)/(x+j+k))
      rs += k * ib / a;                // 1/a - 1/(a+k)
    }
    y = x + (float)n;
  }
  const float z = y + k;
  const float l1 = log1pf(k / y);
  const float iy = 1.0f / y, iz = 1.0f / z, iy2 = iy * iy, iz2 = iz * iz;
  const float Sy = iy * (1.f / 12.f - iy2 * (1.f / 360.f - iy2 * (1.f / 1260.f - iy2 * (1.f / 1680.f))));
  const float Sz = iz * (1.f / 12.f - iz2 * (1.f / 360.f - iz2 * (1.f / 1260.f - iz2 * (1.f / 1680.f))));
  const float Ty = iy2 * (1.f / 12.f - iy2 * (1.f / 120.f - iy2 * (1.f / 252.f - iy2 * (1.f / 240.f))));
  const float Tz = iz2 * (1.f / 12.f - iz2 * (1.f / 120.f - iz2 * (1.f / 252.f - iz2 * (1.f / 240.f))));
  dl = (y - 0.5f) * l1 + k * (logf(z) - 1.0f) + (Sz - Sy) + lp;
  dd = l1 - 0.5f * (iz - iy) - (Tz - Ty) + rs;
}

#define CND(site) ((d.cond >> (site)) & 1u)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// The optimisers fit() accepts (velocity_inference_model.py:76-84,111 hands whatever PyroOptim it is given to SVI):
//   VC_OPT_CLIPPED_ADAM  pyro.optim.ClippedAdam (pyro-ppl 1.8.6 optim/clipped_adam.py; every package tutorial): lr <- lr * lrd before the
//                        update, g = clamp(g, +-clip), [g += wd * p], m / v moments, p -= lr_t sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
//   VC_OPT_ADAM          pyro.optim.Adam = torch.optim.Adam (tutorials/1D_Pancreas_Analysis.ipynb cell 26): [g += wd * p], no clamp, no
//                        decay, p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)   (eps INSIDE the bias correction)
// One element, both of them: p - step_size * m / (sqrt(v) * c2 + eps) with (step_size, c2) of vc_adam_step_size / vc_adam_c2.
// (VC_OPT_CLIPPED_ADAM / VC_OPT_ADAM: include/velocycle_hip.h)
__device__ __forceinline__ float vc_adam_elem(float p, float g, float& m, float& v, float step_size, float fb1, float fb2,
                                              float eps, float clip, float c2 = 1.f, float wd = 0.f) {
  float gi = fminf(fmaxf(g, -clip), clip);
  if (wd != 0.f) gi = gi + wd * p;            // (guarded: 0 * p is NaN for a parameter at -inf, e.g. log of a cov_factor entry clipped to 0)
  m = fb1 * m + (1.f - fb1) * gi;
  v = fb2 * v + (1.f - fb2) * gi * gi;
  return p - step_size * (m / (sqrtf(v) * c2 + eps));
}
// step size of the 1-based optimiser step t; lrd_l, b1l, b2l are natural logs
__device__ __forceinline__ float vc_adam_step_size(long long t, double lr0, double lrd_l, double b1l, double b2l, int kind = VC_OPT_CLIPPED_ADAM) {
  const double td = (double)t;
  if (kind == VC_OPT_ADAM) return (float)(lr0 / (1.0 - exp(td * b1l)));
  return (float)(lr0 * exp(td * lrd_l) * sqrt(1.0 - exp(td * b2l)) / (1.0 - exp(td * b1l)));
}
__device__ __forceinline__ float vc_adam_c2(long long t, double b2l, int kind) {
  return kind == VC_OPT_ADAM ? (float)(1.0 / sqrt(1.0 - exp((double)t * b2l))) : 1.f;
}

// d(-ELBO) / d(unconstrained shape_inv) of one gene (the statements of K_post's / K_tail's shape_inv role): r = 1 / shape_inv,
// U_r = sum_c d loglik / d r from K_main, HDg = the gene's histogram digamma sums, rw = weight of the prior term on this rank
__device__ __forceinline__ float vc_si_grad(const VcDims& d, float r, float si, float U_r, double HDg, float rw) {
  const double lr = (double)logf(r);
  const double dr = (double)U_r + (double)d.nmat_r * d.Nc * (lr + 1.0) + HDg;
  const double gsi = -(double)r * (double)r * dr + (double)rw * ((d.gamma_alpha - 1.f) / si - d.gamma_beta);
  return (float)(-gsi * (double)si);
}

// One wave per histogram TASK (<= 64 distinct count values of one gene and matrix, one per lane), so the
// latency of the kernel is one pass whatever the spread of a gene's counts; K_post adds the few task sums of
// a gene in fixed order.
__device__ __forceinline__ void vc_hist_wave(const VcDims& d, const VcBufs& b, const float* __restrict__ P,
                                             int cond_only, int task, int lane, float si_given = -1.f, int half = 0) {
  const int g = b.h_task[4 * task], m = b.h_task[4 * task + 1];
  const int beg = b.h_task[4 * task + 2], end = b.h_task[4 * task + 3];
  double hl = 0.0, hd = 0.0;
  if ((m == 0 && d.hist_has_S) || (m == 1 && d.hist_has_U)) {
    float si;
    if (CND(VC_SITE_SHAPE_INV)) si = b.cnd[VC_SITE_SHAPE_INV][g];
    else si = si_given > 0.f ? si_given : (cond_only ? 1.f : expf(P[d.poff[VC_P_SHAPE_INV_ULOCS] + g]));
    const float r = 1.0f / si;
    const int i = beg + lane;
    if (i < end) {
      float dl, dd;
      vc_lgamma_digamma_diff(r, b.h_val[i], dl, dd);
      const double n = (double)b.h_cnt[i];
      hl = n * (double)dl;
      hd = n * (double)dd;
    }
    hl = vc_wave_sum_d(hl);
    hd = vc_wave_sum_d(hd);
  }
  if (lane == 0) { b.HL[(size_t)half * b.n_tasks + task] = hl; b.HD[(size_t)half * b.n_tasks + task] = hd; }
}

// Scratch of the dense histogram evaluation ([16 slices][2][64] doubles per matrix in flight): DYNAMIC shared memory, asked for by
// the launches that evaluate dense histograms and by no other -- as a static array it would sit in every kernel that merely
// contains the code (K_omega: 3.6 -> 36 KB per 256-thread block, half the resident blocks; measured + 13 us on the sharded step)
__device__ __forceinline__ double* vc_hist_lds() {
  extern __shared__ double vc_dyn_hist_lds[];
  return vc_dyn_hist_lds;
}
static inline unsigned vc_hist_dyn_lds(const VcDims& d, int with_hist, int nthr) {
  return (with_hist && d.hist_dense) ? (nthr == 1024 ? 4096u : 2048u) * (unsigned)sizeof(double) : 0u;
}

// shape_inv of gene g as a histogram evaluation takes it from the parameters as they stand (the logic of vc_hist_wave)
__device__ __forceinline__ float vc_hist_si(const VcDims& d, const VcBufs& b, const float* __restrict__ P, int cond_only, int g) {
  if (g >= d.Ng) return 1.f;
  if (CND(VC_SITE_SHAPE_INV)) return b.cnd[VC_SITE_SHAPE_INV][g];
  return cond_only ? 1.f : expf(P[d.poff[VC_P_SHAPE_INV_ULOCS] + g]);
}

// Dense form (d.hist_dense): the histogram sums of the 64 genes of gene block `gb` (lane = gene; this thread's gene has
// r = 1 / si) for every matrix the model uses, by ALL `nwv` waves of the block (nwv = 4 or 16; called by every thread of the
// block: barriers inside).  Sixteen slices of the count axis (j = v, v + 16, ...) are summed one after the other in double and
// then added in slice order -- by whichever wave holds them: the result does not depend on the number of waves.
// sm: [16][2][64] doubles.
__device__ __forceinline__ void vc_hist_dense_block(const VcDims& d, const VcBufs& b, int gb, float si, int half, int nwv,
                                                    double* sm /* 2048 doubles */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nblk = d.Ng_pad / 64, nm = d.model == VC_MODEL_VELOCITY ? 2 : 1;
  const int g = gb * 64 + lane;
  const float r = 1.0f / si;
  for (int m = 0; m < nm; ++m) {
    const bool used = (m == 0 && d.hist_has_S) || (m == 1 && d.hist_has_U);
    const int rows = used ? b.hc_rows[m * nblk + gb] : 0;
    const float* __restrict__ tabp = b.HC + (size_t)b.hc_off[m * nblk + gb] * 64 + lane;
    for (int v = wv; v < 16; v += nwv) {
      // eight count levels requested per trip (a plain load-and-add loop is one memory round trip per level); the hardware
      // log2 / reciprocal (v_log_f32, v_rcp_f32: 1 ulp) are what K_main itself uses per element; log2 -> ln once per gene
      double al = 0.0, ad = 0.0;
      constexpr int UB = 8;
      for (int j0 = v; j0 < rows; j0 += 16 * UB) {
        float c[UB];
#pragma unroll
        // (no branch around a load and no select behind it -- hipcc then waits for every single load: a clamped row, dropped below)
        for (int u = 0; u < UB; ++u) { const int j = j0 + 16 * u; c[u] = tabp[(size_t)(j < rows ? j : j0) * 64]; }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const float cu = j0 + 16 * u < rows ? c[u] : 0.f;
          if (j0 + 16 * u >= 256 && __builtin_amdgcn_ballot_w64(cu != 0.f) == 0ull) continue;      // (an empty level: see vc_hist_dense16_finish)
          const float t = r + (float)(j0 + 16 * u);
          al += (double)cu * (double)__builtin_amdgcn_logf(t);
          ad += (double)cu * (double)__builtin_amdgcn_rcpf(t);
        }
      }
      sm[(v * 2 + 0) * 64 + lane] = al;
      sm[(v * 2 + 1) * 64 + lane] = ad;
    }
    __syncthreads();
    if (wv == 0 && g < d.Ng) {
      double hl = 0.0, hd = 0.0;
#pragma unroll
      for (int v = 0; v < 16; ++v) { hl += sm[(v * 2 + 0) * 64 + lane]; hd += sm[(v * 2 + 1) * 64 + lane]; }
      const size_t t = (size_t)half * b.n_tasks + (size_t)nm * g + m;
      b.HL[t] = hl * 0.6931471805599453094;
      b.HD[t] = hd;
    }
    __syncthreads();
  }
}

// The same for a block of exactly 16 waves and ONE matrix (the one-launch tail: a block per (matrix, gene block) -- the evaluation is
// 16 waves of log / rcp / double arithmetic on one CU, 5 us for two matrices, beside idle CUs), in two halves so that the caller can
// put its own work between the request and the use of the table rows: wave v owns slice v of the count axis; `issue` requests its
// first VC_HIST_PRE count levels (nothing is consumed: 640 levels of the table -- a second, dependent round trip for the levels
// beyond 256 cost the blocks that have them 4.5 us at the end of the launch, round 5), `finish` -- once shape_inv of the lane's gene
// is known -- adds them up in increasing order (+ further levels of a block whose largest count exceeds 639), one barrier, and
// wave 0 adds the slices in slice order.  A level beyond 256 at which none of the wave's genes has a cell is skipped by the whole
// wave (its terms are 0 x finite).  Same sums as vc_hist_dense_block, bit for bit.
// QT (round 6): a QUARTER of a gene block by four waves -- 16 genes x 16 slices = 256 threads, lane l of wave w holds gene
// 16 * quarter + (l & 15) and slice 4 w + (l >> 4).  The evaluation is VALU-issue bound (a logarithm, a reciprocal, three float ->
// double conversions and two double FMAs per level), so 16 waves on one CU -- four per SIMD -- take four times what a wave alone
// takes; for the few gene blocks that hold a highly expressed gene (largest count beyond VC_HIST_SPLIT_ROWS) that was 5.6 us at the
// very end of the launch (profiles/r06_hist_split.md).  Those blocks are evaluated by four quarter blocks on four CUs instead, one
// wave per SIMD, each to ITS genes' largest count (hc_rows_q); every (gene, matrix, slice) sum is formed as before and a gene's 16
// slices are added in slice order by one thread -- the same bits.  (All blocks as quarters: measured SLOWER, the launch's wave
// dispatch is serial and 256 blocks of 1024 threads put the gene blocks 4.6 us later.)
#define VC_HIST_PRE 40
#define VC_HIST_SPLIT_ROWS 256
struct VcHistPre { float c[VC_HIST_PRE]; int rows, off, m, gi, v; };
template <bool QT>
__device__ __forceinline__ void vc_hist_dense16_rows(const VcDims& d, const VcBufs& b, int gb, VcHistPre& h, int m, int quarter) {
  const int nblk = d.Ng_pad / 64, nm = d.model == VC_MODEL_VELOCITY ? 2 : 1;
  const bool used = m < nm && ((m == 0 && d.hist_has_S) || (m == 1 && d.hist_has_U));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  h.m = m;
  h.gi = QT ? 16 * quarter + (lane & 15) : lane;
  h.v = QT ? 4 * (wv & 3) + (lane >> 4) : wv;
  h.rows = used ? (QT ? b.hc_rows_q[(m * nblk + gb) * 4 + quarter] : b.hc_rows[m * nblk + gb]) : 0;
  h.off = used ? b.hc_off[m * nblk + gb] : 0;
}
template <bool QT>
__device__ __forceinline__ void vc_hist_dense16_issue(const VcDims& d, const VcBufs& b, VcHistPre& h) {
  const float* __restrict__ tabp = b.HC + (size_t)h.off * 64 + h.gi;
#pragma unroll
  // (no branch around a load and no select behind it: a clamped row, and `finish` drops the value of a level that is not the thread's --
  // with the select HERE hipcc reused one destination register in the quarter form and waited for every single load: 24 round trips,
  // 5.8 us, found on the time stamps)
  for (int u = 0; u < 16; ++u) {
    const int j = h.v + 16 * u, jc = j < h.rows ? j : (h.rows > 0 ? h.rows - 1 : 0);
    h.c[u] = tabp[(size_t)jc * 64];
  }
#pragma unroll
  for (int u = 16; u < VC_HIST_PRE; ++u) h.c[u] = 0.f;
  if (h.rows > 256) {                            // (uniform per block: only a block with counts beyond 255 asks for more)
#pragma unroll
    for (int u = 16; u < VC_HIST_PRE; ++u) {
      const int j = h.v + 16 * u, jc = j < h.rows ? j : h.rows - 1;
      h.c[u] = tabp[(size_t)jc * 64];
    }
  }
}
template <bool QT>
__device__ __forceinline__ void vc_hist_dense16_finish(const VcDims& d, const VcBufs& b, int gb, float si, int half, const VcHistPre& h,
                                                       double* sm /* 2048 doubles */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = QT ? (lane & 15) : lane;
  const int nm = d.model == VC_MODEL_VELOCITY ? 2 : 1;
  const int g = gb * 64 + h.gi;
  const float r = 1.0f / si;
  double al = 0.0, ad = 0.0;
  {
    const float* __restrict__ tabp = b.HC + (size_t)h.off * 64 + h.gi;
#pragma unroll
    for (int u = 0; u < VC_HIST_PRE; ++u) {      // (the association of vc_hist_dense_block: levels in increasing order, one by one)
      const float cu = h.v + 16 * u < h.rows ? h.c[u] : 0.f;        // (what `issue` read for a level beyond the thread's last is a clamped row)
      if (u >= 16 && (h.rows <= 256 || __builtin_amdgcn_ballot_w64(cu != 0.f) == 0ull)) continue;
      const float t = r + (float)(h.v + 16 * u);
      al += (double)cu * (double)__builtin_amdgcn_logf(t);
      ad += (double)cu * (double)__builtin_amdgcn_rcpf(t);
    }
    constexpr int UT = 32;
    // (QT: the lanes of a wave hold four slices -- the wave goes on while any of them has levels left; a lane that is through adds 0 x finite)
    for (int j0 = h.v + 16 * VC_HIST_PRE; QT ? __builtin_amdgcn_ballot_w64(j0 < h.rows) != 0ull : j0 < h.rows; j0 += 16 * UT) {
      float c[UT];
#pragma unroll
      for (int u = 0; u < UT; ++u) { const int j = j0 + 16 * u; c[u] = j < h.rows ? tabp[(size_t)j * 64] : 0.f; }
#pragma unroll
      for (int u = 0; u < UT; ++u) {
        if (__builtin_amdgcn_ballot_w64(c[u] != 0.f) == 0ull) continue;
        const float t = r + (float)(j0 + 16 * u);
        al += (double)c[u] * (double)__builtin_amdgcn_logf(t);
        ad += (double)c[u] * (double)__builtin_amdgcn_rcpf(t);
      }
    }
  }
  sm[(h.v * 2 + 0) * 64 + col] = al;
  sm[(h.v * 2 + 1) * 64 + col] = ad;
  __syncthreads();
  if (wv == 0 && (!QT || lane < 16) && h.m < nm && g < d.Ng) {
    double hl = 0.0, hd = 0.0;
#pragma unroll
    for (int v = 0; v < 16; ++v) { hl += sm[(v * 2 + 0) * 64 + col]; hd += sm[(v * 2 + 1) * 64 + col]; }
    const size_t t = (size_t)half * b.n_tasks + (size_t)nm * g + h.m;
    b.HL[t] = hl * 0.6931471805599453094;
    b.HD[t] = hd;
  }
}

// acc + p[i * ld + j] for i = lane, lane + 64, ... < n, added in that order in double -- eight rows REQUESTED per trip, then the eight adds
// (written as `acc += (double)p[...]` in a loop hipcc waits for every single load: one memory round trip per row; round 6).  Same bits.
__device__ __forceinline__ double vc_col_sum_d(const float* __restrict__ p, int n, int ld, int j, int lane, double acc) {
  for (int i0 = lane; i0 < n; i0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u; v[u] = p[(size_t)(i < n ? i : i0) * ld + j]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) if (i0 + 64 * u < n) acc += (double)v[u];
  }
  return acc;
}

// A 256-thread block of K_pre / K_hist (unfused step, K-particle step): QUARTER `quarter` of gene block `gb`, every matrix the model
// uses, shape_inv from the parameters as they stand.  Round 6: those kernels used to give a gene block ONE 4-wave block (every wave four
// slices of the count axis, one after the other, for both matrices: the block of a highly expressed gene took ~ 9 us -- most of the
// 22 us of the K-particle step's first launch); a quarter block's thread has one slice of one gene.  Same sums (vc_hist_dense16_finish).
__device__ __forceinline__ void vc_hist_dense_quarter(const VcDims& d, const VcBufs& b, int gb, int quarter, const float* __restrict__ P,
                                                      int cond_only, int half, double* sm /* 2048 doubles */) {
  const int nm = d.model == VC_MODEL_VELOCITY ? 2 : 1;
  for (int m = 0; m < nm; ++m) {
    VcHistPre hp;
    vc_hist_dense16_rows<true>(d, b, gb, hp, m, quarter);
    vc_hist_dense16_issue<true>(d, b, hp);
    vc_hist_dense16_finish<true>(d, b, gb, vc_hist_si(d, b, P, cond_only, gb * 64 + hp.gi), half, hp, sm);
    __syncthreads();           // (the next matrix reuses the slices' LDS rows)
  }
}

#endif  // __HIPCC__

// launchers implemented in the .hip translation units -----------------------------------------
// hyper-parameters of the optimiser as the unfused launches take them (kind: VC_OPT_*)
struct VcAdamHyper { double lr0, lrd, b1, b2; float eps, clip, wd; int kind; const unsigned char* frozen; long long frozen_off; };
static inline int vc_hist_blocks(const VcDims& d, const VcBufs& b, int waves) {
  return d.hist_dense ? d.Ng_pad / 64 : (b.n_tasks + waves - 1) / waves;
}
// ... of K_pre / K_hist: dense tables = four quarter blocks per gene block (vc_hist_dense_quarter)
static inline int vc_hist_blocks_pre(const VcDims& d, const VcBufs& b) {
  return d.hist_dense ? (d.Ng_pad / 64) * 4 : (b.n_tasks + 3) / 4;
}
typedef void (*vc_main_launch_fn)(const VcDims& d, const VcBufs& b, hipStream_t st);
vc_main_launch_fn vc_find_main_kernel(int H, int NB, int kind, int noise, int gpl, int c16, const char** name,
                                      const void** kernel);
// run-time-sized kernel set (vc_generic_kernels.hip)
vc_main_launch_fn vc_find_generic_main_kernel(int kind, int noise, const void** kernel);
void vc_launch_pre_generic(const VcDims& d, const VcBufs& b, const float* params, const float* eps, uint64_t seed, long long step,
                           const long long* step_dev, int cond_only, int with_hist, hipStream_t st, int particles, int particle);
#define VC_MAX_PARTICLES 16
struct VcParticleGrads { float* g[VC_MAX_PARTICLES]; int K; };      // gradient buffers of the particles of one step (g[0]: the caller's)
void vc_launch_particle_avg(const VcParticleGrads& pg, long long n, double* loss_ring, long long loss_slots, long long step,
                            long long* step_dev, hipStream_t st);
void vc_launch_pre_particles(const VcDims& d, const VcBufs& b, const VcBufs* bs_dev, const float* params, uint64_t seed,
                             const long long* step_dev, int with_hist, int K, hipStream_t st);
void vc_launch_post_particles(const VcDims& d, const VcBufs& b, const VcBufs* bs_dev, const VcParticleGrads& pg, const float* params,
                              hipStream_t st);
void vc_launch_particle_fin_adam(const VcDims& d, const VcBufs* bs_dev, const VcParticleGrads& pg, float* params, double* loss_dev,
                                 long long loss_slots, long long step, long long* step_dev, double* scratch, float* m, float* v,
                                 const VcAdamHyper& h, int header, long long total, hipStream_t st);
void vc_launch_post_generic(const VcDims& d, const VcBufs& b, const float* params, float* grad, long long* step_dev, hipStream_t st);
void vc_launch_fin_generic(const VcDims& d, const VcBufs& b, const float* params, float* grad, double* loss_dev, long long loss_slots,
                           long long step, const long long* step_dev, hipStream_t st);
void vc_launch_counts_to_u16(const float* src, unsigned short* dst, long long n, hipStream_t st);

void vc_launch_clock_probe(unsigned long long wall_ticks, unsigned long long* out2, hipStream_t st);
// re-layout of one count matrix into [gene block][cell][gbw] (+ log(k+1) for Lognormal noise); with tab != nullptr the
// per-gene count histogram is built on the device in the same pass (vc_small_kernels.hip)
void vc_launch_pack_counts(const float* src, float* dst, long long gene_stride, long long cell_stride,
                           int Ng, int Nc, int nGB, int gbw, int log1p_transform, unsigned* tab, float* ovf_val,
                           int* ovf_gene, unsigned* ovf_n, unsigned ovf_cap, int* bad, const int* ord, hipStream_t st);
void vc_launch_scatter_csr(const long long* indptr, const int* indices, const float* data, float* dst, int Ng, int Nc,
                           int gbw, int log1p_transform, unsigned* tab, float* ovf_val, int* ovf_gene, unsigned* ovf_n,
                           unsigned ovf_cap, int* bad, const int* pos, hipStream_t st);
void vc_launch_expected_logs(const VcDims& d, const VcBufs& b, const float* nu, const float* dnu, const float* phi,
                             const float* omega, const float* logbeta, const float* gamma, float cf_avg, float* out_S,
                             float* out_S2, float* out_U, float* out_U2, hipStream_t st);
void vc_launch_pre(const VcDims& d, const VcBufs& b, const float* params, const float* eps,
                   uint64_t seed, long long step, const long long* step_dev, int cond_only, int with_hist,
                   hipStream_t st, int particles = 1, int particle = 0);
void vc_launch_hist(const VcDims& d, const VcBufs& b, const float* params, int cond_only, hipStream_t st);
void vc_launch_post(const VcDims& d, const VcBufs& b, const float* params, float* grad, long long* step_dev,
                    hipStream_t st);
void vc_launch_fin(const VcDims& d, const VcBufs& b, const float* params, float* grad, double* loss_dev,
                   long long loss_slots, long long step, const long long* step_dev, hipStream_t st);
void vc_launch_fin_adam(const VcDims& d, const VcBufs& b, float* params, float* grad, double* loss_dev,
                        long long loss_slots, long long step, long long* step_dev, float* m, float* v, const VcAdamHyper& h,
                        int header, long long total, hipStream_t st);
// hyper-parameters of pyro's ClippedAdam as the fused kernels take them (logs precomputed on the host)
struct VcAdamArgs {
  float* m;            // exp_avg    [total - header]
  float* v;            // exp_avg_sq [total - header]
  double lr0, lrd_l, b1l, b2l;
  float b1, b2, eps, clip;
  int header;
  float wd;            // weight decay (0: none)
  int kind;            // VC_OPT_*
  const unsigned char* frozen;   // [total] 1 = a parameter tensor without a path to the loss (its site is conditioned): PyroOptim skips it
                                 // (grad None), so it takes no weight decay either; nullptr: none
};
// weight decay of the parameter at flat offset `off` (0 where the tensor is frozen)
__device__ __forceinline__ float vc_wd_at(float wd, const unsigned char* __restrict__ frozen, long long off) {
  return (wd != 0.f && frozen && frozen[off]) ? 0.f : wd;
}

// fused single-rank step (vc_svi_step_fused): K_main(t) -> K_tail(t) -> K_omega(t); boot = 1: sampling only (primes the
// tables for the step *step_dev)
// phase: 0 = the whole single-rank launch, 1 = phase A of the sharded step (writes the exchange buffer `xb`)
void vc_launch_tail(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                    const VcAdamArgs& a, int boot, int phase, const VcXb& xb, hipStream_t st);
// phase B of the sharded step: optimiser on the summed gradient + next sample (gene blocks) and K_omega's blocks, one launch
void vc_launch_phase_b(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                       const VcAdamArgs& a, double* loss_dev, long long loss_slots, int with_hist, const VcXb& xb,
                       hipStream_t st, const VcGate* gate = nullptr);
void vc_launch_tail_x(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                      const VcAdamArgs& a, double* loss_dev, long long loss_slots, int with_hist, const VcXb& xw, const VcXb& xr,
                      const VcGateX& gx, int nc_cap, hipStream_t st);
void vc_launch_omega(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                     const VcAdamArgs& a, double* loss_dev, long long loss_slots, int boot, int with_hist, hipStream_t st);
// tutorial flow on one rank (pw_inline, nothing per cell left to learn): K_tail's gene blocks and K_omega's blocks in one launch
void vc_launch_tail_merged(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                           const VcAdamArgs& a, double* loss_dev, long long loss_slots, hipStream_t st);
// the rest of a single-rank step in ONE launch (round 4): K_tail's gene blocks, its cell blocks with the nu_omega chain inside
// (K_main supplies the partials: pw_inline), the loss block, the histogram blocks (re-deriving the shape_inv update) and the eps
// blocks, side by side
void vc_launch_tail2(const VcDims& d, const VcBufs& b, float* params, float* grad, const long long* step_dev, uint64_t seed,
                     const VcAdamArgs& a, double* loss_dev, long long loss_slots, int with_hist, hipStream_t st);
void vc_launch_p2p_xchg(const VcP2p& p, void* const* regions_dev, long long step, float* out, long long n, long long* status, double timeout_s,
                        unsigned long long* verdict, hipStream_t st);
void vc_launch_adam(float* p, const float* g, float* m, float* v, long long n, const VcAdamHyper& h, long long t_host,
                    const long long* t_dev, const float* loss_hdr, double* loss_ring, long long loss_slots, hipStream_t st);
