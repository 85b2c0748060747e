// ubench.hip — instruction-throughput probes for gfx950 (one number per mode): cycles per
// wave-instruction with W waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#define MFMA4(A, B, C) __builtin_amdgcn_mfma_f32_4x4x1f32((A), (B), (C), 0, 0, 0)

template <int MODE>
__global__ void __launch_bounds__(1024) ub(float* out, int iters) {
  f32x4 acc[8];
  float x[8];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk[8];
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 ba, bb;
  for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)(threadIdx.x * 1e-3f + i); bb[i] = (__bf16)(1.0f + i * 0.01f); }
  const f32x2 pka = {threadIdx.x * 1e-3f, 2e-3f}, pkb = {1.0f + blockIdx.x * 1e-6f, 0.999f};
  const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{a, b, a, b}; x[i] = a + i; pk[i] = f32x2{a + i, a - i}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = MFMA4(a, b, acc[i]);
      if (MODE == 1) acc[i] = MFMA16(a, b, acc[i]);
      if (MODE == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
      if (MODE == 3) x[i] = __builtin_fmaf(x[i], b, a);
      if (MODE == 4) { acc[i] = MFMA4(x[i], b, acc[i]); x[i] = __builtin_amdgcn_exp2f(x[i]); }
      if (MODE == 5) { acc[i] = MFMA16(x[i], b, acc[i]); x[i] = __builtin_amdgcn_exp2f(x[i]); }
      if (MODE == 6) { acc[i] = MFMA4(x[i], b, acc[i]); x[i] = __builtin_fmaf(x[i], b, a); }
      if (MODE == 7) x[i] = __builtin_amdgcn_rcpf(x[i]);
      if (MODE == 12) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0);
      if (MODE == 13) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0); x[i] = __builtin_amdgcn_exp2f(x[i]); }
      if (MODE == 14) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0); x[i] = __builtin_fmaf(x[i], b, a); x[(i + 1) & 7] = __builtin_fmaf(x[(i + 1) & 7], b, a); }
      if (MODE == 15) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0); acc[(i + 4) & 7] = MFMA4(a, b, acc[(i + 4) & 7]); }
      if (MODE == 16) { acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0); x[i] = __builtin_amdgcn_exp2f(x[i]); x[(i + 1) & 7] = __builtin_amdgcn_exp2f(x[(i + 1) & 7]); x[(i + 2) & 7] = __builtin_amdgcn_exp2f(x[(i + 2) & 7]); x[(i + 3) & 7] = __builtin_amdgcn_exp2f(x[(i + 3) & 7]); }
      // round 6: the same bf16 MFMA as a DEPENDENT chain — 17: every MFMA on ONE accumulator, 18: two accumulators alternating,
      // 19: four (the bf16x3 convolutions chain the six piece products of an accumulator)
      if (MODE == 17) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[0], 0, 0, 0);
      if (MODE == 18) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i & 1], 0, 0, 0);
      if (MODE == 19) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i & 3], 0, 0, 0);
      if (MODE == 8) pk[i] = __builtin_elementwise_fma(pk[i], pkb, pka);
      if (MODE == 9) { pk[i] = __builtin_elementwise_fma(pk[i], pkb, pka); acc[i] = MFMA16(a, b, acc[i]); }
      if (MODE == 10) { pk[i] = pk[i] + pka; }
      if (MODE == 11) { acc[i] = MFMA16(a, b, acc[i]); x[i] = __builtin_fmaf(x[i], b, a); x[(i + 1) & 7] = __builtin_fmaf(x[(i + 1) & 7], b, a); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i] + pk[i][0] + pk[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int ub_run(int mode, float* out, int iters, int blocks, int threads, void* st) {
  hipStream_t s = (hipStream_t)st;
  switch (mode) {
    case 0: hipLaunchKernelGGL(ub<0>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 1: hipLaunchKernelGGL(ub<1>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 2: hipLaunchKernelGGL(ub<2>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 3: hipLaunchKernelGGL(ub<3>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 4: hipLaunchKernelGGL(ub<4>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 5: hipLaunchKernelGGL(ub<5>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 6: hipLaunchKernelGGL(ub<6>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 7: hipLaunchKernelGGL(ub<7>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 12: hipLaunchKernelGGL(ub<12>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 13: hipLaunchKernelGGL(ub<13>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 14: hipLaunchKernelGGL(ub<14>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 15: hipLaunchKernelGGL(ub<15>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 16: hipLaunchKernelGGL(ub<16>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 17: hipLaunchKernelGGL(ub<17>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 18: hipLaunchKernelGGL(ub<18>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 19: hipLaunchKernelGGL(ub<19>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 8: hipLaunchKernelGGL(ub<8>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 9: hipLaunchKernelGGL(ub<9>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 10: hipLaunchKernelGGL(ub<10>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 11: hipLaunchKernelGGL(ub<11>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
  }
  return (int)hipGetLastError();
}
