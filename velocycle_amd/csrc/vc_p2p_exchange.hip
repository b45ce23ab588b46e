// One-shot exchange of the sharded step's buffer over peer-mapped device memory ("latency-optimised
// one-shot variant"): instead of an RCCL ring all-reduce of a ~64 KB buffer (latency-bound: 2 (N - 1) hops),
// every rank PUBLISHES its buffer in a region of its own HBM that its peers have mapped (hipIpc), and every rank then reads
// all N buffers directly (xGMI is point to point: N - 1 concurrent reads of 64 KB) and adds them up in FIXED RANK ORDER --
// the same bits on every rank, no float atomics, replicated parameters stay identical.
//
// Region of rank r (one hipMalloc, IPC-exported):  flags[N] (one 64-byte line each) | slot 0 | slot 1     (slots: xb_total floats)
// Step s (engine-local counter, the same on every rank), slot p = s & 1:
//   phase A writes its partials into the rank's own slot p                      (vc_svi_run_sharded points the kernels there)
//   K_xchg, one launch:
//     block 0 publishes: system-scope release, then flags[r] := s + 1 in EVERY rank's region (remote stores)
//     block 0 waits until its own region's flags[q] >= s + 1 for all q (polling LOCAL memory, bounded) and publishes the
//     launch's verdict; the other blocks wait for that verdict
//     system-scope acquire, then  out[i] = sum_{q = 0..N-1} slot_p(q)[i]  in rank order, 16-byte loads that bypass the caches
//   phase B reads `out` (the caller's exchange buffer).
// Two slots suffice: a rank can start writing slot p again (step s + 2) only after its K_xchg(s + 1) has seen every peer's
// flag s + 2, which a peer raises after its phase A(s + 1), i.e. after its reads of step s.
// The kernel boundary in front of K_xchg is what makes phase A's plain stores visible to the peers (end-of-kernel release);
// the flag stores are system-scope atomics, and the region is FINE-GRAINED device memory (vc_p2p_alloc: hipExtMallocWithFlags),
// so that a peer's remote store of a flag is not shadowed by a line the owner's L2 keeps serving while it polls (coarse-grained
// memory is only guaranteed coherent at kernel boundaries).  The wait is BOUNDED (a peer that died must not hang the device): on
// time-out the kernel records it in status[2], SKIPS the sum and poisons the step instead -- NaN into every element of the
// exchange buffer, so that the loss of this step is NaN, the device-side latch fires and the run stops on bad data instead of
// optimising on partial sums -- vc_get_status reports VC_ERR_STATE.
// The verdict is ONE per launch and sticky: block 0 alone waits for the flags and publishes {step, dead} in a word
// every other block of the launch spins on -- never a sum in some blocks and NaN in others; a rank that has poisoned a step
// raises the POISON value in every peer's region (and keeps doing so: status[2] is sticky), so that a peer that was merely
// late to see this rank's flag poisons its own next wait instead of optimising on against a rank that has stopped.
// n is a multiple of 4 by construction (vc_engine.hip rounds the exchange buffer up): no tail elements.
// Correctness across processes is tested with two processes on one device (tests/test_hip_multiproc.py); across xGMI it cannot
// be tested or timed on a 1-GPU box: default stays RCCL, this path is opt-in (VC_EXCHANGE=p2p).
#include "vc_common.h"

#define VC_P2P_FLAG_STRIDE 16      // 64-byte line per flag (in 4-byte words)
#define VC_P2P_POISON 0xFFFFFFFFu  // a rank that gave up on a step publishes this instead of a step number

__global__ __launch_bounds__(256) void vc_p2p_xchg_kernel(VcP2p p, long long step, float* __restrict__ out, long long n,
                                                          long long* __restrict__ status, unsigned long long timeout_ticks,
                                                          unsigned long long* __restrict__ verdict) {
  const unsigned want = (unsigned)(step + 1);
  __shared__ int sm_dead;
  if (threadIdx.x == 0) sm_dead = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    // sticky: once this rank has poisoned a step it never sums again (the caller sees VC_ERR_STATE at its next status check)
    const bool already = status && status[2] != 0;
    if (threadIdx.x < (unsigned)p.world) {
      // publish: this rank's slot is complete (kernel boundary) -> raise flag[rank] in every region, the own one included
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
    if (threadIdx.x == 0)
      __hip_atomic_store(verdict, ((unsigned long long)want << 1) | (unsigned long long)(sm_dead ? 1 : 0), __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_AGENT);
  } else {
    // the launch's one verdict (every block of a <= 64-block launch is resident: block 0 is running)
    if (threadIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned long long v = __hip_atomic_load(verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(v >> 1) == want) { sm_dead = (int)(v & 1ull); break; }
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2 * timeout_ticks + 100000000ull) { sm_dead = 1; break; }
      }
    }
    __syncthreads();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);         // system scope: nothing of the peers' slots may come from a stale line
  const size_t slot_off = (size_t)p.flag_words + (size_t)(step & 1) * (size_t)p.slot_floats;
  const long long n4 = n / 4;
  const bool dead = sm_dead != 0;        // launch-uniform
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dead) {                          // poison instead of a partial sum
      const float qn = __builtin_nanf("");
      reinterpret_cast<float4*>(out)[i] = make_float4(qn, qn, qn, qn);
      continue;
    }
    for (int q = 0; q < p.world; ++q) {
      const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.region[q]) + slot_off) + i;
      float4 v;
      // 16-byte load that bypasses L1 and L2 (sc0 sc1 = system scope): the peer rewrote this slot two steps ago
      asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src) : "memory");
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}

void vc_launch_p2p_xchg(const VcP2p& p, long long step, float* out, long long n, long long* status, double timeout_s,
                        unsigned long long* verdict, hipStream_t st) {
  const long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(vc_p2p_xchg_kernel, dim3(blocks), dim3(256), 0, st, p, step, out, n, status,
                     (unsigned long long)(timeout_s * 1e8), verdict);       // s_memrealtime: 100 MHz
}
