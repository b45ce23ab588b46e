// VALU issue rates on gfx950: cycles per wave64 instruction for fma, pk_fma, exp2, log2, rcp, and mixes, at 1..3 waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
  float a[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);
      if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 1.0001f}, v2f{0.5f, 0.5f});
      if (MODE == 2) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.01f);
      if (MODE == 3) a[i] = __builtin_amdgcn_logf(a[i] + 2.f);
      if (MODE == 4) a[i] = __builtin_amdgcn_rcpf(a[i] + 2.f);
      if (MODE == 5) { a[i] = __builtin_amdgcn_rcpf(a[i]); p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 1.0001f}, v2f{0.5f, 0.5f});
                       p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 1.0001f}, v2f{0.5f, 0.5f}); p[i] = __builtin_elementwise_fma(p[i], v2f{1.0001f, 1.0001f}, v2f{0.5f, 0.5f}); }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> int run(const char* nm, int waves_per_simd, int ninstr_per_iter) {
  float* out; unsigned long long* cyc; CK(hipMalloc(&out, 4 << 20)); CK(hipMalloc(&cyc, 8));
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 100, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per = (double)ms * 1e-3 / ((double)iters * ninstr_per_iter * waves_per_simd);   // seconds per wave-instruction per SIMD
  printf("%-28s waves/SIMD %d: %.2f ns per wave-instr (=%.1f cycles @2.4GHz); s_memtime delta %llu\n", nm, waves_per_simd, per * 1e9, per * 2.4e9, c);
  hipFree(out); hipFree(cyc); return 0;
}
int main() {
  for (int w = 1; w <= 3; ++w) {
    run<0>("v_fma_f32", w, 8); run<1>("v_pk_fma_f32", w, 8); run<2>("v_exp_f32 (+mul)", w, 16);
    run<3>("v_log_f32 (+add)", w, 16); run<4>("v_rcp_f32 (+add)", w, 16); run<5>("1 rcp + 3 pk_fma", w, 32);
  }
  return 0;
}
