// Compile-time specialisations of the small kernels of the fused steps (vc_fused_kernels.hip): the signatures (vc_common.h: VcSig)
// the library carries an instantiation for, and the two halves of the contract --
//   host:   vc_spec_match(d) returns the FIRST row (>= 1) whose signature vc_sig_of(d) satisfies field by field, else 0 (row 0: no facts);
//   device: vc_spec_assume<SPEC>(d) tells the compiler exactly the equalities the host checked (and nothing else) at the top of the kernel.
// A signature not listed here runs the run-time-flag kernels: slower by 2-3 us per launch, otherwise identical (profiles/r05_tail_spec.md).
// Measured: the fields that SELECT code (model, guide, noise, conditioning, kernel kind, histogram form) are ~85 % of the gain, the loop
// bounds the rest -- so the rows of multi-sample data leave the counts of batches and conditions open (VC_SIG_ANY).
// The rows are what `profiles/tools/print_signature.py` prints for the named workload (vc_dbg_signature): not typed by hand.
#pragma once
#include "vc_common.h"

#define VC_SPEC_NONE 0
// kinds of launch a signature's instantiations exist for (a row is compiled only into the kernels its configuration launches)
// (bits: one signature can run under several launch structures -- the phase model's is the same on one rank and on a shard)
#define VC_SPECK_TAIL2 1      // single rank, one launch behind K_main (vc_tail2_kernel)
#define VC_SPECK_MERGED 2     // single rank, tutorial flow (vc_tail_merged_kernel)
#define VC_SPECK_SHARDED 4    // rank of a sharded run (vc_tail_kernel phase A, vc_phase_b_kernel)
#define VC_SPECK_PARTICLES 8  // single rank, Trace_ELBO(num_particles = K) from one call (vc_svi_run_particles: K_pre / K_post of all particles, K_fin + average + optimiser)

struct VcSpecRow { const char* name; int kind; int mq; VcSig sig; };
// clang-format off
static constexpr VcSpecRow VC_SPECS[] = {
  {"generic", 0, 0, {}},
#include "vc_tail_spec_rows.inc"
};
// clang-format on
#define VC_N_SPECS ((int)(sizeof(VC_SPECS) / sizeof(VC_SPECS[0])))

// does the configuration's signature `a` satisfy row `r`?  A field of a row may be VC_SIG_ANY: the row then states nothing about it
// (the kernels keep it a run-time value) -- the "multi" rows leave the number of batches and conditions open that way.
#define VC_SIG_ANY (-1)
static inline bool vc_sig_equal(const VcSig& a, const VcSig& r) {
#define VC_SIG_EQ(f) if (r.f != VC_SIG_ANY && a.f != r.f) return false;
  VC_SIG_FIELDS(VC_SIG_EQ)
#undef VC_SIG_EQ
  return a.cond == r.cond;
}
static inline int vc_spec_match(const VcDims& d) {
  const VcSig s = vc_sig_of(d);
  for (int i = 1; i < VC_N_SPECS; ++i)
    if (vc_sig_equal(s, VC_SPECS[i].sig)) return i;
  return VC_SPEC_NONE;
}

// Launch the instantiation of row `spec` if that row is of launch kind KIND: `launch(mq, sp)` is called with the row's MQ and index as
// std::integral_constant values (compile-time in the callee); false: no such row -- the caller launches the run-time-flag kernel.
// Only rows whose kinds include KIND are ever instantiated by a given caller.
#include <type_traits>
template <int KIND, int I, class F>
static inline bool vc_spec_launch(int spec, F&& launch) {
  if constexpr (I < VC_N_SPECS) {
    if constexpr ((VC_SPECS[I].kind & KIND) != 0) {
      if (spec == I) {
        launch(std::integral_constant<int, VC_SPECS[I].mq>{}, std::integral_constant<int, I>{});
        return true;
      }
    }
    return vc_spec_launch<KIND, I + 1>(spec, launch);
  } else {
    return false;
  }
}

#ifdef __HIPCC__
template <int SPEC>
__device__ __forceinline__ void vc_spec_assume(const VcDims& d) {
  if constexpr (SPEC > 0) {
    constexpr VcSig s = VC_SPECS[SPEC].sig;
#define VC_SIG_ASSUME(f) if constexpr (s.f != VC_SIG_ANY) __builtin_assume(d.f == s.f);
    VC_SIG_FIELDS(VC_SIG_ASSUME)
#undef VC_SIG_ASSUME
    __builtin_assume(d.cond == s.cond);
  }
}
#endif
