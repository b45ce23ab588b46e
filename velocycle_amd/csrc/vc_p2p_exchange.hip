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
// Round 6: by default the exchange has NO launch of its own any more -- phase B runs vc_p2p_gate (this protocol) in its own blocks and
// adds the ranks' slots where it reads them (vc_xget, vc_common.h; vc_tuning.p2p_separate = 1 keeps K_xchg): K_main, phase A, phase B.
// Correctness across processes is tested with two processes on one device (tests/test_hip_multiproc.py); across xGMI it cannot
// be tested or timed on a 1-GPU box: default stays RCCL, this path is opt-in (VC_EXCHANGE=p2p).
#include "vc_common.h"

// (the publish / wait protocol itself is vc_p2p_gate, vc_common.h: phase B runs the same gate when the exchange is folded into it)
__global__ __launch_bounds__(256) void vc_p2p_xchg_kernel(VcP2p p, void* const* __restrict__ regions, long long step,
                                                          float* __restrict__ out, long long n, long long* __restrict__ status,
                                                          unsigned long long timeout_ticks, unsigned long long* __restrict__ verdict) {
  const bool dead = vc_p2p_gate(regions, p.world, p.rank, step, status, timeout_ticks, verdict) != 0;        // launch-uniform
  const size_t slot_off = (size_t)p.flag_words + (size_t)(step & 1) * (size_t)p.slot_floats;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dead) {                          // poison instead of a partial sum
      const float qn = __builtin_nanf("");
      reinterpret_cast<float4*>(out)[i] = make_float4(qn, qn, qn, qn);
      continue;
    }
    // rank order; the four dwords of two ranks requested together (round 6: one rank's load used to be awaited before the next
    // was issued -- N dependent remote round trips)
    for (int q0 = 0; q0 < p.world; q0 += 2) {
      float v[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float* src = vc_xslot(regions, q0 + u < p.world ? q0 + u : q0) + slot_off + 4 * i;      // (the table through the scalar cache: vc_xget)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[u][k] = vc_xload(src + k);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (q0 + u < p.world) { acc.x += v[u][0]; acc.y += v[u][1]; acc.z += v[u][2]; acc.w += v[u][3]; }
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}

void vc_launch_p2p_xchg(const VcP2p& p, void* const* regions_dev, long long step, float* out, long long n, long long* status,
                        double timeout_s, unsigned long long* verdict, hipStream_t st) {
  const long long n4 = n / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(vc_p2p_xchg_kernel, dim3(blocks), dim3(256), 0, st, p, regions_dev, step, out, n, status,
                     (unsigned long long)(timeout_s * 1e8), verdict);       // s_memrealtime: 100 MHz
}
