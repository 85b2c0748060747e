// lds_atomic_ubench.hip — cost of LDS float atomics on gfx950 (question behind the fused attention
// backward's dQ accumulation): cycles per wave-instruction per CU for ds_add_f32 with 64 distinct
// addresses, 4 lanes per address, ds_add_u32, plain ds_write_b32 / read-modify-write for comparison.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -munsafe-fp-atomics
#include <hip/hip_runtime.h>

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  __shared__ float buf[16 * 1024];
  for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) buf[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // MODE 0/2/3/4: 64 distinct consecutive addresses per wave; MODE 1: 16 addresses x 4 lanes
  float* p = buf + wave * 256 + (MODE == 1 ? (lane & 15) : lane);
  unsigned int* pu = reinterpret_cast<unsigned int*>(p);
  float v = 1.0f + lane * 1e-3f, acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* q = p + j * 64;  // four different rows per iteration
      if (MODE == 0 || MODE == 1) atomicAdd(q, v);
      if (MODE == 2) atomicAdd(reinterpret_cast<unsigned int*>(q), (unsigned int)lane);
      if (MODE == 3) { *(volatile float*)q = v; }
      if (MODE == 4) { float t = *(volatile float*)q; *(volatile float*)q = t + v; }
      if (MODE == 5) acc += atomicAdd(q, v);  // returning
    }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = buf[threadIdx.x] + acc + (float)pu[0];
}

extern "C" int run(int mode, float* out, int iters, int blocks, int threads, void* st) {
  hipStream_t s = (hipStream_t)st;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, s, out, iters); break;
  }
  return (int)hipGetLastError();
}
