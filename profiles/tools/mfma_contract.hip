// north_star allows MFMA "only for the dense harmonics x coeff contraction".  This tool settles with a NUMBER whether it pays
// (VERDICT r2 item 7): the contraction of the likelihood kernel's cell loop,
//     eta_S[g,c] = nu0_g + cf_c + ns_g sin(phi_c) + nc_g cos(phi_c)       dd[g,c] = ns_g cos(phi_c) - nc_g sin(phi_c)
// (e2 = -(eta_S - nu0 - cf) costs nothing extra on either path), for ONE TILE of 16 cells x 64 genes = 1024 (gene, cell) pairs,
//   V  as the kernel does it: v_pk_fma_f32 chains on gene pairs (lane = 2 genes of 128, 8 cells per tile): 6 packed ops per pair
//      and cell = 48 v_pk_* per tile and wave;
//   M  on the matrix pipe: v_mfma_f32_16x16x4_f32 with K padded to 4 -- A = [nu0 ns nc 1] (16 genes x 4), B = [1 sin cos cf]^T
//      (4 x 16 cells) gives eta_S, A' = [ns nc 0 0], B' = [cos -sin 0 0]^T gives dd: 2 MFMA per 16 genes x 16 cells = 8 per tile;
// both leave 32 result registers per lane (same registers out), which are then consumed by the SAME block of follow-up VALU
// work standing in for the rest of the cell body (REST packed ops + TR transcendentals per tile; REST = TR = 0: the contraction
// alone).  Operands stay in registers, no memory traffic, every CU busy, W waves per SIMD (256-thread workgroups, W per CU).
// Reported: shader cycles per tile and wave (s_memtime of one wave) and ns per tile and SIMD (hipEvents).
//   hipcc -O3 --offload-arch=gfx950 profiles/tools/mfma_contract.hip -o scratch/mfma_contract && ./scratch/mfma_contract
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void pkfma(v2f& a, v2f b, v2f c) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c)); }
__device__ __forceinline__ void pkmul(v2f& a, v2f b, v2f c) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); }
__device__ __forceinline__ void pkadd(v2f& a, v2f b, v2f c) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c)); }
__device__ __forceinline__ void exp1(float& a) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); }

// follow-up work on the 32 result registers: REST packed fmas and TR transcendentals, spread evenly (stands in for the
// observation model + gradient accumulation of the cell body; identical for both variants)
template <int REST, int TR>
__device__ __forceinline__ void rest_of_cell(v2f (&r)[16], v2f k0, v2f k1) {
  constexpr int NT = TR > 0 ? TR : 1;
  int done = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (TR > 0) { float x = r[t % 16].x; exp1(x); r[t % 16].x = x; }
    const int upto = (REST * (t + 1)) / NT;
#pragma unroll
    for (int i = done; i < upto; ++i) pkfma(r[(i * 5 + 3) % 16], k0, k1);
    done = upto;
  }
}

// MODE 0: VALU contraction, MODE 1: MFMA contraction
template <int MODE, int REST, int TR>
__global__ __launch_bounds__(256) void k(float* out, int tiles, unsigned long long* ticks) {
  const int lane = threadIdx.x & 63;
  // per-gene operands: VALU: pairs {nu0, ns, nc} of the lane's 2 genes; MFMA: one A register per M tile (4 x 16 genes) and matrix
  v2f nu0 = v2f{0.3f + lane * 1e-3f, 0.4f}, ns = v2f{0.1f, 0.2f + lane * 1e-3f}, nc = v2f{-0.2f, 0.15f};
  float a_eta[4], a_dd[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { a_eta[m] = 0.1f * (lane % 16) + 0.01f * m + (lane / 16); a_dd[m] = 0.05f * (lane % 16) - 0.02f * m; }
  v2f acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = v2f{0.f, 0.f};
  const v2f k0 = v2f{1.0001f, 0.9999f}, k1 = v2f{1e-3f, -1e-3f};
  // per-cell operands of the tile: VALU: {sin, cos, cf} of 8 cells as wave-uniform pairs; MFMA: one B register per matrix
  v2f sn[8], cs[8], cf[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { sn[c] = v2f{0.1f * c, 0.1f * c}; cs[c] = v2f{1.f - 0.05f * c, 1.f - 0.05f * c}; cf[c] = v2f{0.01f * c, 0.01f * c}; }
  float b_eta = 0.3f + 0.01f * lane, b_dd = 0.7f - 0.01f * lane;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < tiles; ++it) {
    v2f r[16];                 // 32 result registers: eta_S and dd of the tile
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        v2f t, es, dd;
        pkmul(t, nc, cs[c]); pkfma(t, ns, sn[c]);          // t = ns sin + nc cos
        pkadd(es, nu0, cf[c]); pkadd(es, es, t);           // eta_S = nu0 + cf + t
        pkmul(dd, nc, sn[c]); asm volatile("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(dd) : "v"(ns), "v"(cs[c]));
        r[2 * c] = es; r[2 * c + 1] = dd;
      }
    } else {
      const v4f z = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const v4f e = __builtin_amdgcn_mfma_f32_16x16x4f32(a_eta[m], b_eta, z, 0, 0, 0);
        const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dd[m], b_dd, z, 0, 0, 0);
        r[4 * m] = v2f{e.x, e.y}; r[4 * m + 1] = v2f{e.z, e.w}; r[4 * m + 2] = v2f{d.x, d.y}; r[4 * m + 3] = v2f{d.z, d.w};
      }
    }
    rest_of_cell<REST, TR>(r, k0, k1);
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(r[i]));     // consume (both variants)
    asm volatile("" : "+v"(b_eta), "+v"(b_dd), "+v"(nu0), "+v"(ns));      // operands opaque per tile: nothing is hoisted
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

template <int MODE, int REST, int TR>
int run(const char* what, int w, float* out, unsigned long long* ticks) {
  const int tiles = 20000, ncu = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, REST, TR>), dim3(ncu * w), dim3(256), 0, 0, out, tiles / 10, ticks);      // warm-up
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, REST, TR>), dim3(ncu * w), dim3(256), 0, 0, out, tiles, ticks);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long t = 0;
  CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
  // per tile and SIMD: the SIMD's w waves each did `tiles` tiles
  printf("{\"what\": \"%s\", \"contraction\": \"%s\", \"rest_pk\": %d, \"rest_trans\": %d, \"waves_per_simd\": %d, \"cycles_per_tile_per_wave\": %.1f, "
         "\"ns_per_tile_per_simd\": %.2f}\n", what, MODE ? "mfma_f32_16x16x4" : "v_pk_fma_f32", REST, TR, w, (double)t / tiles,
         ms * 1e6 / ((double)tiles * w));
  return 0;
}

int main() {
  float* out; unsigned long long* ticks;
  CK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CK(hipMalloc(&ticks, 8));
  for (int w : {1, 2}) {
    // the contraction alone
    if (run<0, 0, 0>("contraction only", w, out, ticks)) return 1;
    if (run<1, 0, 0>("contraction only", w, out, ticks)) return 1;
    // with the rest of the S+U cell body beside it: per tile of 1024 pairs = 8 wave-instructions per packed op of the body;
    // the body has 43 packed + 12 transcendental per gene pair and cell, 6 of the packed are the contraction:
    // (43 - 6) x 8 = 296 packed and 12 x 8 = 96 transcendental instructions follow the contraction
    if (run<0, 296, 96>("with the rest of the S+U cell body", w, out, ticks)) return 1;
    if (run<1, 296, 96>("with the rest of the S+U cell body", w, out, ticks)) return 1;
  }
  return 0;
}
