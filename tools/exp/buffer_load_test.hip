// Pins the raw-buffer-load idiom on gfx950 before it goes into the convolution kernels' staging loads:
//   address = descriptor base + voffset (VGPR, bytes) + soffset (SGPR, bytes); reads beyond num_records return 0.
// Measured on MI355X (round 4): the gather is exact incl. 242 out-of-bounds zeros; (a) voffset 0x7ffffff0 + soffset 64 -> 0: the sum is
// range-checked without wrapping, so a huge voffset is a safe "no load" encoding; (b) voffset -4 -> 0: voffset is UNSIGNED, a negative
// lane offset is out of range, not a signed sum — put the slack into the base; (c) 16-byte loads at offsets that are only 4-byte
// aligned return wrong data (192 of 256 values): b128 buffer loads need 16-byte aligned addresses; (d) stores beyond num_records are
// dropped, the last valid word is written.
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

// Probes for the follow-up forms: (a) a huge voffset (0x7ffffff0) with a non-zero soffset — is the sum range-checked without
// wrapping; (b) voffset = -4; (c) a 16-byte load at a 4-byte aligned offset; (d) a store beyond num_records is dropped.
__global__ void probe_kernel(float* buf, float* out, unsigned bytes) {
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(buf, bytes);
  typedef int i32x4_ __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x;
  const int so = __builtin_amdgcn_readfirstlane(64);
  out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 0x7ffffff0, so, 0));          // (a) want 0
  out[64 + lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, -4, so, 0));              // (b) want 0 (or buf[15])
  const i32x4_ q = __builtin_amdgcn_raw_buffer_load_b128(rs, 4 + 16 * lane, 0, 0);                               // (c) buf[1 + 4 lane ..]
  out[128 + 4 * lane + 0] = __builtin_bit_cast(float, q[0]); out[128 + 4 * lane + 1] = __builtin_bit_cast(float, q[1]);
  out[128 + 4 * lane + 2] = __builtin_bit_cast(float, q[2]); out[128 + 4 * lane + 3] = __builtin_bit_cast(float, q[3]);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, -5.0f), rs, (int)bytes + 4 * lane, 0, 0);        // (d) dropped
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, 9.0f), rs, (int)bytes - 4, 0, 0);               // last valid word
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
  {
    const int nb = 2048;  // floats in the probed buffer; the allocation is twice that so that a stray store is visible
    std::vector<float> hb(2 * nb);
    for (int i = 0; i < 2 * nb; ++i) hb[i] = 100.0f + i;
    float *db, *dout2;
    CK(hipMalloc(&db, 2 * nb * 4)); CK(hipMalloc(&dout2, 512 * 4));
    CK(hipMemcpy(db, hb.data(), 2 * nb * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, db, dout2, (unsigned)(nb * 4));
    CK(hipDeviceSynchronize());
    std::vector<float> po(512), after(2 * nb);
    CK(hipMemcpy(po.data(), dout2, 512 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(after.data(), db, 2 * nb * 4, hipMemcpyDeviceToHost));
    printf("(a) voffset 0x7ffffff0 + soffset 64 -> %g (0 = range check without wrap)\n", po[0]);
    printf("(b) voffset -4 + soffset 64 -> %g (0 = out of range; %g = buf[15], i.e. a signed sum)\n", po[64], hb[15]);
    int bad128 = 0;
    for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) bad128 += po[128 + 4 * l + c] != hb[1 + 4 * l + c];
    printf("(c) 16-byte loads at 4-byte aligned offsets: %d of 256 values wrong\n", bad128);
    int stray = 0;
    for (int i = nb; i < 2 * nb; ++i) stray += after[i] != hb[i];
    printf("(d) stores beyond num_records: %d words changed (0 = dropped); last valid word = %g (9 = stored)\n", stray, after[nb - 1]);
  }
  return bad != 0;
}
