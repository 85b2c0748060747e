// Pins the raw-buffer-load idiom on gfx950 before it goes into the convolution kernels' staging loads:
//   address = descriptor base + voffset (VGPR, bytes) + soffset (SGPR, bytes); reads beyond num_records return 0.
// hipcc --offload-arch=gfx950 -O2 tools/exp/buffer_load_test.hip -o tools/exp/buffer_load_test.bin && tools/exp/buffer_load_test.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// descriptor: base, stride 0, num_records = bytes, flags = dword 3 (gfx90a / gfx94x / gfx950 raw buffer: DATA_FORMAT = 32)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// out[c][i] = in[c * plane + idx[i]] for c < C through ONE descriptor, per-channel offset in soffset
__global__ void gather_kernel(const float* in, const int* idx, float* out, int n, int C, int plane, unsigned in_bytes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, in_bytes);
  const int vo = (i < n ? idx[i] : 0) * 4;
  for (int c = 0; c < C; ++c) {
    const int so = __builtin_amdgcn_readfirstlane(c * plane * 4);
    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, so, 0));
    if (i < n) out[(size_t)c * n + i] = v;
  }
}

int main() {
  const int plane = 1024, C = 8, n = 4096;
  std::vector<float> h((size_t)C * plane);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0f + (float)i;
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = (i * 37) % (plane + 64);  // the last 64 indices of a plane run past it: into the
                                                                 // next channel, and for channel C-1 out of bounds
  float *din, *dout; int* didx;
  CK(hipMalloc(&din, h.size() * 4)); CK(hipMalloc(&dout, (size_t)C * n * 4)); CK(hipMalloc(&didx, n * 4));
  CK(hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(didx, idx.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(gather_kernel, dim3(n / 256), dim3(256), 0, 0, din, didx, dout, n, C, plane, (unsigned)(h.size() * 4));
  CK(hipDeviceSynchronize());
  std::vector<float> o((size_t)C * n);
  CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0, oob = 0;
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < n; ++i) {
      const size_t e = (size_t)c * plane + idx[i];
      const float want = e < h.size() ? h[e] : 0.f;  // beyond num_records: 0
      oob += e >= h.size();
      if (o[(size_t)c * n + i] != want) { if (bad < 5) printf("c %d i %d: got %g want %g\n", c, i, o[(size_t)c * n + i], want); ++bad; }
    }
  printf("%d of %d values wrong (%d of them were out-of-bounds reads expected to return 0)\n", bad, C * n, oob);
  printf(bad ? "FAILED\n" : "raw buffer loads: base + voffset + soffset and the out-of-bounds clamp behave as assumed\n");
  return bad != 0;
}
