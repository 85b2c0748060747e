// Does a hipMemsetAsync / hipMemset2DAsync node captured into a hipGraph hit the right bytes when the destination lies far
// (> 4 GB) inside ONE large allocation? (tests/guard serves captured allocations from a 32 GB arena.)
// hipcc --offload-arch=gfx950 -O2 tools/exp/memset_graph_test.hip -o /tmp/memset_graph_test && /tmp/memset_graph_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(float* p, float v, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
int main() {
  const size_t GB = 1ull << 30;
  char* arena = nullptr;
  CK(hipMalloc((void**)&arena, 12 * GB));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t offs[] = {0, 1 * GB + 4096, 3 * GB + 512, 4 * GB + 1024, 5 * GB + 8192, 9 * GB + 256};
  const size_t n = 3136;  // floats: dq of one image, 4 channels x 784
  int bad_total = 0;
  for (int two_d = 0; two_d < 2; ++two_d)
    for (size_t off : offs) {
      float* p = reinterpret_cast<float*>(arena + off);
      hipLaunchKernelGGL(fill, dim3((n * 12 + 255) / 256), dim3(256), 0, st, p, 7.0f, n * 12);
      CK(hipStreamSynchronize(st));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      if (two_d) CK(hipMemset2DAsync(p, 40 * 784 * 4, 0, n * 4, 1, st));
      else CK(hipMemsetAsync(p, 0, n * 4, st));
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 3; ++rep) {  // later launches of the SAME executable graph, destination dirtied between
        if (rep) { hipLaunchKernelGGL(fill, dim3((n * 12 + 255) / 256), dim3(256), 0, st, p, 7.0f, n * 12); }
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        if (rep < 2) {
          std::vector<float> h2(n);
          CK(hipMemcpy(h2.data(), p, n * 4, hipMemcpyDeviceToHost));
          int nz2 = 0; for (size_t i = 0; i < n; ++i) nz2 += h2[i] != 0.f;
          if (nz2) printf("  launch %d: %d floats not zero (first %g)\n", rep, nz2, h2[0]);
        }
      }
      std::vector<float> h(n * 2);
      CK(hipMemcpy(h.data(), p, n * 2 * 4, hipMemcpyDeviceToHost));
      int nz = 0, keep = 0;
      for (size_t i = 0; i < n; ++i) nz += h[i] != 0.f;
      for (size_t i = n; i < 2 * n; ++i) keep += h[i] == 7.0f;
      printf("%s offset %5.2f GB: %d of %zu floats NOT zeroed, %d of %zu neighbours intact\n", two_d ? "memset2D" : "memset  ",
             off / (double)GB, nz, n, keep, n);
      bad_total += nz + (int)(n - keep);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  printf(bad_total ? "FAILED\n" : "all graph memsets hit their bytes\n");
  return bad_total != 0;
}
