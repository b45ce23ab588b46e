// VALU issue costs on gfx950 at 1..8 waves per SIMD (256-thread workgroups = one wave per SIMD each, w of them per CU, every
// CU busy): v_fma_f32, v_pk_fma_f32, v_exp_f32, v_log_f32, v_rcp_f32 each alone (16 independent chains per wave), and the
// instruction MIX of the likelihood kernels' cell loops (per gene pair, from profiles/valu_model.json: S+U 39 packed : 6
// plain : 8 exp/log : 4 rcp; U-only 21 : 5 : 6 : 2; S-only 16 : 5 : 4 : 2 -- after round 3's arithmetic diet) with the transcendentals spread between the packed
// operations as in the kernels.  Reported per wave64 instruction and SIMD: ns (hipEvents) and shader-clock ticks (s_memtime of
// one wave), so that bench.py can price a kernel at the clock its own probe reads.  No memory traffic, no cross-lane work:
// this is the arithmetic floor of the mix, the bound `roofline.valu` uses.
//   hipcc -O3 --offload-arch=gfx950 profiles/tools/valu_rate.hip -o scratch/valu_rate 2>/dev/null && ./scratch/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
constexpr int NCH = 16;
// inline asm: the compiler would otherwise pair adjacent scalar fmas into v_pk_fma_f32 (SLP) and change the mix
__device__ __forceinline__ void fma1(float& a, float b, float c) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
__device__ __forceinline__ void pkfma(v2f& a, v2f b, v2f c) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
__device__ __forceinline__ void exp1(float& a) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); }
__device__ __forceinline__ void log1(float& a) { asm volatile("v_log_f32 %0, %0" : "+v"(a)); }
__device__ __forceinline__ void rcp1(float& a) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); }

// MODE 0..4: one instruction class; MODE 5: a mix of PK packed, PL plain, EL exp/log and RC rcp per gene pair, 4 pairs per trip
template <int MODE, int PK, int PL, int EL, int RC>
__global__ void k(float* out, int iters, unsigned long long* ticks) {
  float a[NCH]; v2f p[NCH];
  for (int i = 0; i < NCH; ++i) { a[i] = 1.0f + threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
  const v2f c1 = v2f{1.0001f, 0.9999f}, c0 = v2f{0.5f, 0.25f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE <= 4) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        if (MODE == 0) fma1(a[i], c1.x, c0.x);
        if (MODE == 1) pkfma(p[i], c1, c0);
        if (MODE == 2) exp1(a[i]);       // the value saturates; the issue cost does not depend on it
        if (MODE == 3) log1(a[i]);
        if (MODE == 4) rcp1(a[i]);
      }
    } else {
      constexpr int NT = EL + RC;                       // transcendentals per pair, one every PK / NT packed operations
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int done = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t < EL) { if (t & 1) log1(a[(4 * q + t) % NCH]); else exp1(a[(4 * q + t) % NCH]); }
          else rcp1(a[(4 * q + t) % NCH]);
          const int upto = (PK * (t + 1)) / NT;
#pragma unroll
          for (int i = done; i < upto; ++i) pkfma(p[(5 * q + i) % NCH], c1, c0);
          done = upto;
        }
#pragma unroll
        for (int j = 0; j < PL; ++j) fma1(a[(4 * q + NT + j) % NCH], c1.x, c0.x);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < NCH; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}
template <int MODE, int PK, int PL, int EL, int RC> int run(const char* nm, int waves_per_simd) {
  float* out; unsigned long long* ticks;
  CK(hipMalloc(&out, 8 << 20)); CK(hipMalloc(&ticks, 8));
  const int ninstr = MODE == 5 ? 4 * (PK + PL + EL + RC) : NCH;
  const int iters = 8000000 / ninstr;                   // ~20-30 ms per launch: the clock has settled
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  dim3 grid(256 * waves_per_simd), block(256);
  hipLaunchKernelGGL((k<MODE, PK, PL, EL, RC>), grid, block, 0, 0, out, iters, ticks);      // warm-up, same length
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, PK, PL, EL, RC>), grid, block, 0, 0, out, iters, ticks);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long t; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
  const double n = (double)iters * ninstr * waves_per_simd;         // wave-instructions per SIMD
  const double ns = (double)ms * 1e6 / n, tk = (double)t / n;
  printf("{\"what\": \"%s\", \"waves_per_simd\": %d, \"ns_per_instr\": %.3f, \"ticks_per_instr\": %.3f, \"ticks_per_ns\": %.3f",
         nm, waves_per_simd, ns, tk, (double)t / (ms * 1e6));
  if (MODE == 5) printf(", \"instr_per_cell_iter\": %d, \"ns_per_cell_iter\": %.1f, \"ticks_per_cell_iter\": %.1f", ninstr, ns * ninstr, tk * ninstr);
  printf("}\n");
  CK(hipFree(out)); CK(hipFree(ticks)); return 0;
}
int main() {
  for (int w : {1, 2, 3, 4, 6, 8}) {
    run<0, 0, 0, 0, 0>("v_fma_f32", w); run<1, 0, 0, 0, 0>("v_pk_fma_f32", w); run<2, 0, 0, 0, 0>("v_exp_f32", w);
    run<3, 0, 0, 0, 0>("v_log_f32", w); run<4, 0, 0, 0, 0>("v_rcp_f32", w);
    run<5, 39, 6, 8, 4>("mix vfull (S+U)", w); run<5, 21, 5, 6, 2>("mix vu (U only)", w); run<5, 16, 5, 4, 2>("mix phase (S only)", w);
  }
  return 0;
}
