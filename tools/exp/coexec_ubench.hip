// coexec_ubench.hip — do plain VALU instructions of ONE wave run under the bf16 MFMAs of ANOTHER wave of the same SIMD?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (role A) and 4-7 (role B) share the four SIMDs pairwise.
//   mode 0: A = MFMA stream, B exits          mode 1: A exits, B = VALU stream (v_and / v_sub / v_perm mix)
//   mode 2: A = MFMA, B = VALU (the question) mode 3: both roles run MFMA + VALU interleaved in blocks (24 MFMA | 48 VALU)
//   mode 4: as 3, role B rotated by half a period (starts with its VALU block)
//   mode 5: as 2 with s_setprio 3 on role A   mode 6: as 3 with s_setprio 1 around the MFMA blocks
//   mode 7: as 3, role B at s_setprio 1 throughout   mode 8: as 3, role B at s_setprio 3   mode 9: as 3, priority alternates per iteration
// out[wave] = cycles (s_memtime) of that wave's loop. Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ void mfma_block(f32x4 (&acc)[8], const bf16x8& a, const bf16x8& b) {
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = MF(a, b, acc[i]);
}
__device__ __forceinline__ void valu_block(unsigned (&x)[8], unsigned k) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // 3 dependent plain VALU ops per element: and, sub, perm
      unsigned t = x[i] & 0xffff0000u;
      t = x[i] - t + k;
      x[i] = __builtin_amdgcn_perm(t, x[(i + 1) & 7], 0x07060302u);
    }
}

__global__ void __launch_bounds__(512) coexec(long long* out, int iters, int mode) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool roleA = wave < 4;
  f32x4 acc[8];
  unsigned x[8];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 1.f, 2.f, 3.f}; x[i] = threadIdx.x * 2654435761u + i; }
  const unsigned k = blockIdx.x + 1;
  if ((mode == 0 && !roleA) || (mode == 1 && roleA)) return;
  if (mode == 5 && roleA) __builtin_amdgcn_s_setprio(3);
  if (mode == 7 && !roleA) __builtin_amdgcn_s_setprio(1);
  if (mode == 8 && !roleA) __builtin_amdgcn_s_setprio(3);
  const long long t0 = clock64();
  if (mode <= 2 || mode == 5) {
    if (roleA) for (int it = 0; it < iters; ++it) mfma_block(acc, a, b);
    else for (int it = 0; it < iters; ++it) valu_block(x, k);
  } else {
    if (mode == 4 && !roleA) valu_block(x, k);
    for (int it = 0; it < iters; ++it) {
      if (mode == 9) { if (((it & 1) != 0) == roleA) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
      if (mode == 6) __builtin_amdgcn_s_setprio(1);
      mfma_block(acc, a, b);
      if (mode == 6) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      valu_block(x, k);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + (float)x[i];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + (s == 12345.678f ? 1 : 0);
}

extern "C" int coexec_run(long long* out, int iters, int mode, int blocks, void* st) {
  hipLaunchKernelGGL(coexec, dim3(blocks), dim3(512), 0, (hipStream_t)st, out, iters, mode);
  return (int)hipGetLastError();
}
