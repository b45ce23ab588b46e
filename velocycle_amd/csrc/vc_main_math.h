// Building blocks of the likelihood kernels (vc_main_kernel.h: the compiled fast set; vc_generic_kernels.hip: the run-time-sized
// set): the cell record as scalar-loaded SGPR pairs, packed-pair float32 arithmetic, the observation models, and the count loads
// issued from inline asm with hand-placed waits.
#pragma once
#include "vc_common.h"

typedef float v2f __attribute__((ext_vector_type(2)));

// Cell record as stored in the cell table: every value duplicated {x, x}, so that a scalar load
// delivers it as an SGPR pair that v_pk_*_f32 consume directly as a packed operand (no per-cell
// v_mov splat).  Layout (pairs): sin k, cos k (k = 1..H), Db[0..NB), omega, cf, then (velocity model, VC_OMEGA_CS)
// k omega cos k, k omega sin k (k = 1..H).  With VC_FOLD_LOG2E and a count noise model the record's omega is omega * ln 2
// and its cf is cf * log2 e (vc_common.h: vc_rec_*_scale): eta comes out in log2 units without a multiply.
template <int H, int NB>
struct VcCellRec {
  v2f sn[H], cs[H];
  v2f db[NB > 0 ? NB : 1];
  v2f omega, cf;
  v2f ocs[H], osn[H];       // S+U kernel only
};

// The record is read through the constant address space: the table is written by K_pre, never by this
// kernel, and only a constant-space load of a wave-uniform address is selected as s_load_dwordx8 (scalar
// cache, SGPR pairs as packed operands) instead of a 64-lane vector load of one address.
template <int H, int NB, bool XT>
__device__ __forceinline__ VcCellRec<H, NB> vc_load_cell(const float* __restrict__ ct) {
  typedef const __attribute__((address_space(4))) v2f* cptr;
  cptr c2 = (cptr)(const void*)ct;
  VcCellRec<H, NB> r;
#pragma unroll
  for (int k = 0; k < H; ++k) { r.sn[k] = c2[2 * k]; r.cs[k] = c2[2 * k + 1]; }
#pragma unroll
  for (int q = 0; q < NB; ++q) r.db[q] = c2[2 * H + q];
  r.omega = c2[2 * H + NB];
  r.cf = c2[2 * H + NB + 1];
  if (XT) {
#pragma unroll
    for (int k = 0; k < H; ++k) { r.ocs[k] = c2[2 * H + NB + 2 + 2 * k]; r.osn[k] = c2[2 * H + NB + 3 + 2 * k]; }
  }
  return r;
}

// The kernel's partial sums (GO: one row per workgroup and output, written by every workgroup in its last microsecond; CO: per-cell
// sums) are read by the NEXT launch.  Left dirty in the eight L2s they are written back at the kernel boundary, which then
// costs their bytes / 6 TB/s on top of the boundary itself (MI355X_MICROARCH.md, "boundary": 2.8-3.8 us behind 12.6-16.8 MB of
// fp32 partials).  VC_WT_STORES=1: they go out as write-through stores (system-scope relaxed atomic store = global_store ... sc0 sc1),
// so that the write-back overlaps the workgroups that are still computing.
// Measured (profiles/r04_two_launch.md): the step gains 0.5 % (V-joint 50k x 2k), the kernel itself reads 2-3 us longer because the
// write-back now happens inside it -- off by default.
#ifndef VC_WT_STORES
#define VC_WT_STORES 0
#endif
__device__ __forceinline__ void vc_store_out(float* p, float v) {
#if VC_WT_STORES
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  *p = v;
#endif
}

// ---------------------------------------------------------------------------------------------
// Packed-pair arithmetic.  The kernel is VALU-issue bound before it is HBM bound (rocprof: VALU busy
// ~100 %, 4 cycles per wave64 instruction, 8 per transcendental), and v_pk_{fma,mul,add}_f32 retire two
// genes per issue slot, so the per-element math is written on float2 pairs (4 genes/lane = 2 pairs).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ v2f v2(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f v2_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// hardware base-2 transcendentals; arguments are never denormal here (t = r + mu >= r > 0, zp >= 1e-5)
__device__ __forceinline__ v2f v2_exp2(v2f x) { return v2f{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ v2f v2_log2(v2f x) { return v2f{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
__device__ __forceinline__ v2f v2_rcp(v2f x) { return v2f{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }


// Observation model of a pair of counts k with mean mu (= exp of the log-mean, eta2 = log-mean*log2 e):
//   a   = d loglik / d eta (natural units)
//   ll += k (eta2 - log2 t)   [NB, t = r + mu]  |  k eta2 - mu log2 e  [Poisson]    (log2 units)
//   lt += log2 t              [NB]: sum_c log(r+mu) enters the loss (times r) and d/dr.
// Nothing else of the NB needs per-element work: sum_c (r+k)/(r+mu) = n + (sum_c a)/r, and the r-only
// terms (r log r, lgamma) come from the per-gene count histograms (K_pre / K_post).
// NOLOSS (gradient-only instantiations, vc_set_loss_every): the terms that enter the loss VALUE only are not formed
template <int NOISE, bool NOLOSS = false>
__device__ __forceinline__ void vc_obs_counts(v2f k, v2f eta2, v2f mu, v2f r, v2f& a, v2f& ll, v2f& lt) {
  if (NOISE == VC_NOISE_NB) {
    const v2f t = r + mu;
    const v2f lt2 = v2_log2(t);
    const v2f it = v2_rcp(t);
    a = (r * (k - mu)) * it;
    if (!NOLOSS) ll = v2_fma(k, eta2 - lt2, ll);
    lt += lt2;
  } else {
    a = k - mu;
    ll = v2_fma(k, eta2, ll) - mu * VC_LOG2E;
  }
}
// Lognormal: y = log(count + 1) ~ Normal(eta, s); ll carried in natural units / ln 2 to share the rescale
__device__ __forceinline__ void vc_obs_lognormal(v2f y, v2f eta, float inv_s2, v2f& a, v2f& ll) {
  const v2f e = y - eta;
  a = e * inv_s2;
  ll = v2_fma(e * (-0.5f * VC_LOG2E), a, ll);
}

// ---------------------------------------------------------------------------------------------
// Count loads with hand-placed waits (VC_ASM_LOADS).  An asm load is invisible to hipcc's s_waitcnt bookkeeping
// (cdna_hip_programming.md section 5.7): the destination tuple counts as written at the end of the statement, so
//   * every consumer sits behind a wait statement that names the tuple "+v" (pins the order), and
//   * a drain statement naming every tuple closes the loop (a load landing after the registers were re-used would corrupt
//     the epilogue), and
//   * tests/test_tools_cpu.py checks the code object: no instruction touches a tuple between its load and its wait, no scratch.
// Address form: SGPR base (wave-uniform row of the blocked layout) + 32-bit VGPR lane offset; `s_nop 4` covers a base that was
// produced by v_readfirstlane (VALU write of an SGPR -> VMEM read: 5 wait states).
// ---------------------------------------------------------------------------------------------
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));

template <int NDW> struct VcCnt;                  // the count registers of one (cell, matrix) of a lane: NDW dwords
template <> struct VcCnt<2> { v2u a; };
template <> struct VcCnt<4> { v4u a; };
template <> struct VcCnt<8> { v4u a, b; };

// VC_NT_LOADS: the counts are read exactly once per step -- the S+U kernel marks its loads non-temporal (streaming: no reuse).
// Measured (2 000 genes, same box, profiles/r04_two_launch.md): the 8-genes-per-lane S+U kernel at 50 000 cells 118.3 -> 117.2 us and the
// whole step 137.2 -> 132.1 us (the small launch behind it finds more of what it reads still cached); the 4-genes-per-lane S+U
// kernel (shards of 6 250 ... 25 000 cells) 21.9 / 34.3 / 59.3 -> 22.8 / 35.3 / 61.4 us; the U-only kernel unchanged; the S-only (phase)
// kernel 55.3 -> 65.4 us -- so only the 8-genes-per-lane S+U kernel marks its loads.
#ifndef VC_NT_LOADS
#define VC_NT_LOADS 1
#endif
template <bool NT> __device__ __forceinline__ void vc_issue(VcCnt<2>& c, uint32_t voff, const char* sbase) {
  if (NT && VC_NT_LOADS) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 nt" : "=&v"(c.a) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=&v"(c.a) : "v"(voff), "s"(sbase) : "memory");
}
template <bool NT> __device__ __forceinline__ void vc_issue(VcCnt<4>& c, uint32_t voff, const char* sbase) {
  if (NT && VC_NT_LOADS) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=&v"(c.a) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(c.a) : "v"(voff), "s"(sbase) : "memory");
}
template <bool NT> __device__ __forceinline__ void vc_issue(VcCnt<8>& c, uint32_t voff, const char* sbase) {
  if (NT && VC_NT_LOADS)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3 nt\n\tglobal_load_dwordx4 %1, %2, %3 offset:16 nt"
                 : "=&v"(c.a), "=&v"(c.b) : "v"(voff), "s"(sbase) : "memory");
  else
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:16"
                 : "=&v"(c.a), "=&v"(c.b) : "v"(voff), "s"(sbase) : "memory");
}
// wait until at most N vector-memory operations of this wave are outstanding; the tuples named are readable afterwards
template <int N> __device__ __forceinline__ void vc_wait(VcCnt<2>& c) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(c.a) : "n"(N)); }
template <int N> __device__ __forceinline__ void vc_wait(VcCnt<4>& c) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(c.a) : "n"(N)); }
template <int N> __device__ __forceinline__ void vc_wait(VcCnt<8>& c) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(c.a), "+v"(c.b) : "n"(N)); }
template <int N> __device__ __forceinline__ void vc_wait(VcCnt<2>& c, VcCnt<2>& e) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(c.a), "+v"(e.a) : "n"(N)); }
template <int N> __device__ __forceinline__ void vc_wait(VcCnt<4>& c, VcCnt<4>& e) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(c.a), "+v"(e.a) : "n"(N)); }
template <int N> __device__ __forceinline__ void vc_wait(VcCnt<8>& c, VcCnt<8>& e) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(c.a), "+v"(c.b), "+v"(e.a), "+v"(e.b) : "n"(N));
}
template <int NDW> __device__ __forceinline__ uint32_t vc_cnt_dword(const VcCnt<NDW>& c, int k);
template <> __device__ __forceinline__ uint32_t vc_cnt_dword<2>(const VcCnt<2>& c, int k) { return c.a[k]; }
template <> __device__ __forceinline__ uint32_t vc_cnt_dword<4>(const VcCnt<4>& c, int k) { return c.a[k]; }
template <> __device__ __forceinline__ uint32_t vc_cnt_dword<8>(const VcCnt<8>& c, int k) { return k < 4 ? c.a[k] : c.b[k - 4]; }

