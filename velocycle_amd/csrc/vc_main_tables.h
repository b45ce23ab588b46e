// Launcher and tables of the likelihood kernel's instantiations (included at the end of vc_main_kernel.h): one table of 30
// entries (H in 1..3, NB in 0..4, 4 / 8 genes per lane) per (kind, noise, count storage [, gradient-only]) and translation unit.
#pragma once
template <int H, int NB, int KIND, int NOISE, int GPL, int C16>
static void vc_main_launch(const VcDims& d, const VcBufs& b, hipStream_t st) {
  hipLaunchKernelGGL((vc_main_kernel<H, NB, KIND, NOISE, GPL, C16>), dim3(d.n_main_wg), dim3(256), vc_main_dyn_lds(d), st, d, b);
}

struct VcMainEntry { int H, NB, kind, noise, gpl, c16; vc_main_launch_fn fn; const void* kernel; };

// Explicit kernel instantiations are needed in both compilation passes; the launcher table is host-only.
// The uint16 variants exist for the count noise models only (Lognormal stores log(k + 1)).
#define VC_INST_1(KIND, NOISE, H, NB, GPL, C16) \
  template __global__ void vc_main_kernel<H, NB, KIND, NOISE, GPL, C16>(const VcDims, const VcBufs);
#define VC_ENT_1(KIND, NOISE, H, NB, GPL, C16)                                   \
  {H, NB, KIND, NOISE, GPL, C16, &vc_main_launch<H, NB, KIND, NOISE, GPL, C16>, \
   (const void*)&vc_main_kernel<H, NB, KIND, NOISE, GPL, C16>},
#define VC_FOR_NB(M, KIND, NOISE, H, GPL, C16)                                                                  \
  M(KIND, NOISE, H, 0, GPL, C16) M(KIND, NOISE, H, 1, GPL, C16) M(KIND, NOISE, H, 2, GPL, C16) M(KIND, NOISE, H, 3, GPL, C16) \
  M(KIND, NOISE, H, 4, GPL, C16)
#define VC_FOR_H(M, KIND, NOISE, GPL, C16) \
  VC_FOR_NB(M, KIND, NOISE, 1, GPL, C16) VC_FOR_NB(M, KIND, NOISE, 2, GPL, C16) VC_FOR_NB(M, KIND, NOISE, 3, GPL, C16)
#define VC_FOR_ALL_CS(M, KIND, NOISE, CS) VC_FOR_H(M, KIND, NOISE, 4, CS) VC_FOR_H(M, KIND, NOISE, 8, CS)
#define VC_FOR_ALL_F32(M, KIND, NOISE) VC_FOR_ALL_CS(M, KIND, NOISE, 0)
#define VC_FOR_ALL_U16(M, KIND, NOISE) VC_FOR_ALL_CS(M, KIND, NOISE, 1)

#if defined(__HIP_DEVICE_COMPILE__)
#define VC_DEFINE_TABLE(NAME, KIND, NOISE) VC_FOR_ALL_F32(VC_INST_1, KIND, NOISE)
#define VC_DEFINE_TABLE_U16(NAME, KIND, NOISE) VC_FOR_ALL_U16(VC_INST_1, KIND, NOISE)
#define VC_DEFINE_TABLE_CS(NAME, KIND, NOISE, CS) VC_FOR_ALL_CS(VC_INST_1, KIND, NOISE, CS)
#else
#define VC_DEFINE_TABLE(NAME, KIND, NOISE)      \
  VC_FOR_ALL_F32(VC_INST_1, KIND, NOISE)        \
  extern const VcMainEntry NAME[30] = {VC_FOR_ALL_F32(VC_ENT_1, KIND, NOISE)};
#define VC_DEFINE_TABLE_U16(NAME, KIND, NOISE)  \
  VC_FOR_ALL_U16(VC_INST_1, KIND, NOISE)        \
  extern const VcMainEntry NAME[30] = {VC_FOR_ALL_U16(VC_ENT_1, KIND, NOISE)};
#define VC_DEFINE_TABLE_CS(NAME, KIND, NOISE, CS)  \
  VC_FOR_ALL_CS(VC_INST_1, KIND, NOISE, CS)        \
  extern const VcMainEntry NAME[30] = {VC_FOR_ALL_CS(VC_ENT_1, KIND, NOISE, CS)};
#endif
