// mlp_fused.hip — the transformer block's position-wise MLP as ONE forward and ONE backward kernel.
//
// Reference: TransformerBlock._out = Sequential(Conv2d(C, 4C, 1), GELU(), Conv2d(4C, C, 1)) and
// `x + self._out(self._ln2(x))` (models/autoregressive/image_gpt.py:43-52), C = 16 for the
// BASELINE.json configuration. Unfused, the 4C-channel hidden tensor (205 MB at batch 1024) crosses
// HBM eight times per layer (written by fc1, read by fc2; in backward written by fc2's data gradient,
// read + rewritten by GELU', read by fc1's data and weight gradients, and read again (GELU
// recomputed) by fc2's weight gradient): 545 us per layer, bandwidth bound. Here it never leaves
// the register file: the hidden activations of a 16-pixel tile are one wave's accumulator registers.
//
// All products run on v_mfma_f32_16x16x4_f32 (exact fp32) with pixels on the N axis:
//   H^T  [64 x px] = W1 [64 x 16] X^T [16 x px]       A = W1 fragments (VGPR, loaded once per wave)
//   G    = gelu(H + b1)                               D layout: lane (px j, g), VGPR (m, r) <-> hidden
//   Y^T  [16 x px] = W2 [16 x 64] G^T [64 x px]           unit 16m + 4g + r — which is exactly a valid
//                                                     B fragment for K-step (m, r): no data movement
// and in backward
//   dG^T [64 x px] = W2^T dY^T ;  dH = dG * gelu'(H)      (same D layout as H: elementwise in registers)
//   dX^T [16 x px] = W1^T dH^T                            (B fragment = the dH registers again)
//   dW2 += dY^T G ,  dW1 += dH^T X                        contraction over the 16 pixels: G and dH go
//                                                     through a 5 KB LDS transpose per wave, dY and X
//                                                     are re-read from L2 in the transposed order
// Weight/bias gradients are accumulated in registers over all the tiles of a wave, reduced across the
// workgroup's waves in LDS, written as one partial row per workgroup and summed by a second kernel
// (deterministic, no atomics), exactly like conv_wgrad.hip.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

constexpr int C = 16, HD = 64;           // channels, hidden units (the only instantiated shape)
constexpr int MLP_THREADS = 256;         // 4 waves
constexpr int PART = HD * C + HD + C * HD + C;  // dW1 | db1 | dW2 | db2 floats per partial row

__device__ __forceinline__ float gelu_f(float x) { return pg_gelu(x); }

struct MlpArgs {
  const float* x; const float* w1; const float* b1; const float* w2; const float* b2;
  const float* res; const float* dy;
  float* y; float* dx; float* part;
  int N, L, tiles_per_img, total_tiles;
};

// ------------------------------------------------------------------------------------ forward
__global__ void __launch_bounds__(MLP_THREADS) mlp_fwd_kernel(const MlpArgs a) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * (MLP_THREADS / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (MLP_THREADS / 64);

  // weight fragments, loaded once: W1 as A[i = hidden 16m+j][k = channel 4s+g];
  // W2 as A[i = out channel j][k = hidden 16m+4g+r] for K-step (m, r)
  float w1f[4][4], w2f[4][4];
  f32x4 b1r[4], b2r;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      w1f[m][s] = a.w1[(16 * m + j) * C + 4 * s + g];
      w2f[m][s] = a.w2[j * HD + 16 * m + 4 * g + s];
      b1r[m][s] = a.b1[16 * m + 4 * g + s];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) b2r[r] = a.b2[4 * g + r];

  for (int tile = wave; tile < a.total_tiles; tile += nwaves) {
    const int n = tile / a.tiles_per_img;
    const int p = (tile - n * a.tiles_per_img) * 16 + j;
    const size_t base = (size_t)n * C * a.L + p;
    float xb[4];
    f32x4 rv;
#pragma unroll
    for (int s = 0; s < 4; ++s) xb[s] = a.x[base + (size_t)(4 * s + g) * a.L];
#pragma unroll
    for (int r = 0; r < 4; ++r) rv[r] = a.res ? a.res[base + (size_t)(4 * g + r) * a.L] : 0.f;
    f32x4 h[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      h[m] = b1r[m];
#pragma unroll
      for (int s = 0; s < 4; ++s) h[m] = MFMA16(w1f[m][s], xb[s], h[m]);
    }
    f32x4 yv = b2r + rv;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) yv = MFMA16(w2f[m][r], gelu_f(h[m][r]), yv);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) a.y[base + (size_t)(4 * g + r) * a.L] = yv[r];
  }
}

// ------------------------------------------------------------------------------------ backward
constexpr int TS = 20;  // LDS row stride (floats) of the transposed [hidden][16 px] tiles: 16-byte
                        // aligned rows, write banks hidden*20 + px distinct over a 32-lane group

__global__ void __launch_bounds__(MLP_THREADS) mlp_bwd_kernel(const MlpArgs a) {
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (MLP_THREADS / 64) + wv;
  const int nwaves = gridDim.x * (MLP_THREADS / 64);
  float* tg = lds + (size_t)wv * 2 * HD * TS;  // G^T  [64][TS]
  float* th = tg + HD * TS;                     // dH^T [64][TS]

  // W1 as A[i = hidden 16m+j][k = channel 4s+g]         (H recompute)
  // W2^T as A[i = hidden 16m+j][k = out channel 4s+g]   (dG = W2^T dY)
  // W1^T as A[i = channel j][k = hidden 16m+4g+r]       (dX = W1^T dH), K-step (m, r)
  float w1f[4][4], w2t[4][4], w1t[4][4];
  f32x4 b1r[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      w1f[m][s] = a.w1[(16 * m + j) * C + 4 * s + g];
      w2t[m][s] = a.w2[(4 * s + g) * HD + 16 * m + j];
      w1t[m][s] = a.w1[(16 * m + 4 * g + s) * C + j];
      b1r[m][s] = a.b1[16 * m + 4 * g + s];
    }
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[4], acc2[4], db1[4];  // dW1 [hidden 16m+4g+r][c = j]; dW2 [co = 4g+r][hidden 16m+j]
  float db2[4];                    // db2 partial for out channel 4s+g, this lane's pixel
#pragma unroll
  for (int m = 0; m < 4; ++m) { acc1[m] = zero4; acc2[m] = zero4; db1[m] = zero4; db2[m] = 0.f; }

  for (int tile = wave; tile < a.total_tiles; tile += nwaves) {
    const int n = tile / a.tiles_per_img;
    const int p0 = (tile - n * a.tiles_per_img) * 16;
    const size_t img = (size_t)n * C * a.L;
    const size_t base = img + p0 + j;
    float xb[4], dyb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xb[s] = a.x[base + (size_t)(4 * s + g) * a.L];    // B[k = channel 4s+g][px j]
      dyb[s] = a.dy[base + (size_t)(4 * s + g) * a.L];
    }
    // the same tiles in the transposed fragment order (channel j, pixels 4g..4g+3): L2 hits
    const f32x4 xt = *reinterpret_cast<const f32x4*>(a.x + img + (size_t)j * a.L + p0 + 4 * g);
    const f32x4 dyt = *reinterpret_cast<const f32x4*>(a.dy + img + (size_t)j * a.L + p0 + 4 * g);

    f32x4 h[4], dg[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      h[m] = b1r[m];
      dg[m] = zero4;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        h[m] = MFMA16(w1f[m][s], xb[s], h[m]);
        dg[m] = MFMA16(w2t[m][s], dyb[s], dg[m]);
      }
    }
    // G = gelu(H), dH = dG * gelu'(H); both also go to LDS transposed for the weight gradients
    f32x4 dxv = zero4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = h[m][r];
        float cdf, ee;
        pg_gelu_parts(hv, cdf, ee);
        const float pdf = 0.39894228040143267794f * ee;
        const float gv = hv * cdf;
        const float dh = dg[m][r] * (cdf + hv * pdf);
        const int hid = 16 * m + 4 * g + r;
        tg[hid * TS + j] = gv;
        th[hid * TS + j] = dh;
        db1[m][r] += dh;
        dxv = MFMA16(w1t[m][r], dh, dxv);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) a.dx[base + (size_t)(4 * g + r) * a.L] = dxv[r];
#pragma unroll
    for (int s = 0; s < 4; ++s) db2[s] += dyb[s];
    // weight gradients: contraction over the tile's 16 pixels, pixel 4k+e in K-step e of lane group k
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 gt = *reinterpret_cast<const f32x4*>(tg + (16 * m + j) * TS + 4 * g);  // B[k=px][j=hidden]
      const f32x4 ht = *reinterpret_cast<const f32x4*>(th + (16 * m + j) * TS + 4 * g);  // A[i=hidden][k=px]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc2[m] = MFMA16(dyt[e], gt[e], acc2[m]);  // dW2[co][hidden] += dY[px][co] G[px][hidden]
        acc1[m] = MFMA16(ht[e], xt[e], acc1[m]);   // dW1[hidden][c] += dH[px][hidden] X[px][c]
      }
    }
  }

  // ---- reduce the workgroup's four waves through LDS, one partial row per workgroup
  __syncthreads();
  float* red = lds;  // [4 waves][PART]
  float* mine = red + (size_t)wv * PART;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[(16 * m + 4 * g + r) * C + j] = acc1[m][r];                    // dW1[hidden][c]
      mine[HD * C + HD + (4 * g + r) * HD + 16 * m + j] = acc2[m][r];     // dW2[co][hidden]
      // bias gradients: sum over the 16 pixel lanes of the group
      float v = db1[m][r];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (j == 0) mine[HD * C + 16 * m + 4 * g + r] = v;
    }
    float w = db2[m];
    w += __shfl_xor(w, 1, 64); w += __shfl_xor(w, 2, 64); w += __shfl_xor(w, 4, 64); w += __shfl_xor(w, 8, 64);
    if (j == 0) mine[HD * C + HD + C * HD + 4 * m + g] = w;
  }
  __syncthreads();
  float* prow = a.part + (size_t)blockIdx.x * PART;
  for (int i = threadIdx.x; i < PART; i += MLP_THREADS)
    prow[i] = (red[i] + red[PART + i]) + (red[2 * PART + i] + red[3 * PART + i]);
}

// out[i] += sum_rows part[row][i], routed to the four gradient tensors
__global__ void __launch_bounds__(256) mlp_reduce_kernel(const float* __restrict__ part, int rows,
                                                         float* __restrict__ dw1, float* __restrict__ db1,
                                                         float* __restrict__ dw2, float* __restrict__ db2) {
  __shared__ float red[32][9];
  const int sl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int s = blockIdx.x * 8 + sl;
  float a0 = 0.f, a1 = 0.f;
  if (s < PART) {
    const float* p = part + s;
    int r = rg;
    for (; r + 32 < rows; r += 64) {
      a0 += p[(size_t)r * PART];
      a1 += p[(size_t)(r + 32) * PART];
    }
    if (r < rows) a0 += p[(size_t)r * PART];
  }
  red[rg][sl] = a0 + a1;
  __syncthreads();
  if (rg != 0 || s >= PART) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) acc += red[r][sl];
  if (s < HD * C) dw1[s] += acc;
  else if (s < HD * C + HD) db1[s - HD * C] += acc;
  else if (s < HD * C + HD + C * HD) dw2[s - HD * C - HD] += acc;
  else db2[s - HD * C - HD - C * HD] += acc;
}

int mlp_bwd_blocks(int N, int L) {
  const long tiles = (long)N * (L / 16);
  long b = (tiles + 4 * 8 - 1) / (4 * 8);  // >= 8 tiles per wave
  if (b > 512) b = 512;
  return b < 1 ? 1 : (int)b;
}

int check_shape(const char* who, int N, int Cc, int Hd, int L) {
  PG_REQUIRE(N > 0 && L > 0, PG_EINVAL, "%s: non-positive dimension", who);
  PG_REQUIRE(Cc == C && Hd == HD, PG_ESHAPE, "%s: only C=16, hidden=64 is instantiated (got %d, %d)", who, Cc, Hd);
  PG_REQUIRE(L % 16 == 0, PG_ESHAPE, "%s: L=%d is not a multiple of 16", who, L);
  return 0;
}

}  // namespace

PG_EXPORT size_t pg_mlp_gelu_bwd_workspace_floats(int N, int L) {
  if (N <= 0 || L < 16) return 0;
  return (size_t)mlp_bwd_blocks(N, L) * PART;
}

PG_EXPORT int pg_mlp_gelu_fwd(const float* x, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* res, float* y, int N, int Cc, int Hd,
                              int L, void* stream) {
  PG_REQUIRE(x && w1 && b1 && w2 && b2 && y, PG_EINVAL, "pg_mlp_gelu_fwd: null pointer");
  int rc = check_shape("pg_mlp_gelu_fwd", N, Cc, Hd, L);
  if (rc) return rc;
  MlpArgs a = {};
  a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.res = res; a.y = y;
  a.N = N; a.L = L; a.tiles_per_img = L / 16; a.total_tiles = N * a.tiles_per_img;
  long blocks = (a.total_tiles + 4 * 4 - 1) / (4 * 4);  // >= 4 tiles per wave
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3((unsigned)blocks), dim3(MLP_THREADS), 0, (hipStream_t)stream, a);
  PG_LAUNCH_CHECK("pg_mlp_gelu_fwd");
  return 0;
}

PG_EXPORT int pg_mlp_gelu_bwd(const float* x, const float* w1, const float* b1, const float* w2,
                              const float* dy, float* dx, float* dw1, float* db1, float* dw2,
                              float* db2, int N, int Cc, int Hd, int L, float* workspace,
                              size_t workspace_floats, void* stream) {
  PG_REQUIRE(x && w1 && b1 && w2 && dy && dx && dw1 && db1 && dw2 && db2 && workspace, PG_EINVAL,
             "pg_mlp_gelu_bwd: null pointer");
  int rc = check_shape("pg_mlp_gelu_bwd", N, Cc, Hd, L);
  if (rc) return rc;
  PG_REQUIRE(workspace_floats >= pg_mlp_gelu_bwd_workspace_floats(N, L), PG_EINVAL,
             "pg_mlp_gelu_bwd: workspace too small");
  PG_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0, PG_EINVAL,
             "pg_mlp_gelu_bwd: x / dy must be 16-byte aligned");
  MlpArgs a = {};
  a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.dy = dy; a.dx = dx; a.part = workspace;
  a.N = N; a.L = L; a.tiles_per_img = L / 16; a.total_tiles = N * a.tiles_per_img;
  const int blocks = mlp_bwd_blocks(N, L);
  hipStream_t st = (hipStream_t)stream;
  const size_t tr = (size_t)4 * 2 * HD * TS, rd = (size_t)4 * PART;
  const size_t shmem = (tr > rd ? tr : rd) * sizeof(float);
  hipLaunchKernelGGL(mlp_bwd_kernel, dim3((unsigned)blocks), dim3(MLP_THREADS), shmem, st, a);
  PG_LAUNCH_CHECK("pg_mlp_gelu_bwd");
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)((PART + 7) / 8)), dim3(256), 0, st, workspace,
                     blocks, dw1, db1, dw2, db2);
  PG_LAUNCH_CHECK("pg_mlp_gelu_bwd(reduce)");
  return 0;
}
