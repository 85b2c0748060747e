// conv_wgrad_small.hip — weight / bias gradient of convolutions with VERY FEW input channels and many taps:
// the models' input layers (PixelCNN 7x7 on 1 channel, PixelSNAIL / ImageGPT 3x3 on 3 / 1 channels:
// reference pixel_cnn.py:86-92, pixel_snail.py:152-158, image_gpt.py:88-94; the weight output of
// aten::convolution_backward behind loss.backward(), trainer.py:180; unmasked taps, nn/convolution.py:42).
//
//   dw[co][ci][t] = sum_{n,r,c} dy[n,co,r,c] * act(x[n,ci,r+dr_t,c+dc_t])      db[co] = sum dy
//
// conv_wgrad.hip tiles this GEMM over input CHANNELS, which leaves the matrix cores idle when there is one
// channel: PixelCNN's 7x7 layer took 1.47 ms per step (profiles/README.md, round 3 item 6). Here the GEMM
// is M = co (64 per workgroup), N = the (ci, tap) pairs + one "ones" column for the bias (<= 64 columns),
// K = pixels: v_mfma_f32_16x16x4_f32 with
//   A[i = co][k]   = dy[co][pixel 4 kk + k]                                (LDS tile [co][pixels])
//   B[k][j = col]  = x[ci_j][pixel 4 kk + k shifted by tap_j]              (LDS tile with halo; per-lane base
//                                                                           offset of column j + uniform pixel offset)
// A wave owns one 16-co tile and all four column tiles: per K step 1 + 4 fragment reads (ds_read_b32) and
// 4 MFMAs. Persistent workgroups over (image, row tile), one row of partial sums per workgroup, then
// conv_wgrad.hip's deterministic reduction.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

constexpr int SW_THREADS = 256;
constexpr int SW_CO = 64;
constexpr int SW_COLS = 64;

struct SwArgs {
  const float* x; const float* dy; float* part;
  long part_stride;
  int N, Cin, Cout, H, W, T;
  int TR, tiles_per_img, total_tiles;
  int min_dr, min_dc, tile_w, xrows;   // x tile: xrows x tile_w per channel, origin (row0 + min_dr, min_dc)
  int dstride;                         // dy tile row stride (floats), TR * W + pad
  int ncols, bias_col;                 // used columns (Cin * T [+ 1]); index of the ones column or -1
  int in_act;
  int col_off[SW_COLS];                // LDS float offset of column j inside the x tile (ci plane + tap shift); -1: unused
};

__global__ void __launch_bounds__(SW_THREADS) conv_wgrad_small_kernel(const SwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int co0 = blockIdx.y * SW_CO;
  const int TP = a.TR * a.W;             // tile pixels (multiple of 4)
  float* dyt = lds;                      // [64][dstride]
  float* xt = lds + SW_CO * a.dstride;   // [Cin][xrows][tile_w]
  const int xplane = a.xrows * a.tile_w;
  const int i = lane & 15, kq = lane >> 4;

  int boff[4];   // per column tile: this lane's base offset into xt (+ kq), or -1 (zero) / -2 (one)
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int j = 16 * n + i;
    boff[n] = j == a.bias_col ? -2 : (j < a.ncols && a.col_off[j] >= 0 ? a.col_off[j] + kq : -1);
  }
  f32x4 acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int W4 = a.W >> 2;
  const int co_valid = min(SW_CO, a.Cout - co0);

  for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
    const int n_img = tile / a.tiles_per_img;
    const int row0 = (tile - n_img * a.tiles_per_img) * a.TR;
    const int rows = min(a.TR, a.H - row0);
    __syncthreads();  // the previous tile's fragment reads are done
    // ---- dy tile: float4 rows (W % 4 == 0), zero beyond the image / the valid channels
    {
      const float* dyb = a.dy + ((size_t)n_img * a.Cout + co0) * a.H * a.W + (size_t)row0 * a.W;
      const int q_per_co = TP >> 2;
      for (int e = tid; e < SW_CO * q_per_co; e += SW_THREADS) {
        const int co = e / q_per_co, q = e - co * q_per_co;
        const int p = 4 * q, r = p / a.W;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co < co_valid && r < rows) v = *reinterpret_cast<const float4*>(dyb + (size_t)co * a.H * a.W + p);
        *reinterpret_cast<float4*>(dyt + co * a.dstride + p) = v;
      }
    }
    // ---- x tile with halo, activation applied, zero outside the image
    {
      const float* xb = a.x + (size_t)n_img * a.Cin * a.H * a.W;
      for (int e = tid; e < a.Cin * xplane; e += SW_THREADS) {
        const int ci = e / xplane, rem = e - ci * xplane;
        const int tr = rem / a.tile_w, tc = rem - tr * a.tile_w;
        const int ir = row0 + a.min_dr + tr, ic = a.min_dc + tc;
        float v = 0.f;
        if (ir >= 0 && ir < a.H && ic >= 0 && ic < a.W) v = pg_apply_act(xb[((size_t)ci * a.H + ir) * a.W + ic], a.in_act);
        xt[e] = v;
      }
    }
    __syncthreads();
    // ---- K loop: 4 pixels (of one image row) per step
    const float* ap = dyt + (16 * wave + i) * a.dstride + kq;
    const int ksteps = (rows * a.W) >> 2;
    for (int kk = 0; kk < ksteps; ++kk) {
      const int r = kk / W4, c4 = kk - r * W4;
      const float av = ap[4 * kk];
      const int xo = r * a.tile_w + 4 * c4;   // uniform: pixel (r, 4 c4) of the tile in x-tile coordinates (before the tap shift)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float bv = boff[n] >= 0 ? xt[boff[n] + xo] : (boff[n] == -2 ? 1.f : 0.f);
        acc[n] = MFMA16(av, bv, acc[n]);
      }
    }
  }
  // ---- partial sums: D[row = co 16 wave + 4 kq + r][col = 16 n + i]
  float* prow = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int j = 16 * n + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * wave + 4 * kq + r;
      if (co >= a.Cout) continue;
      if (j == a.bias_col) prow[(size_t)a.Cout * a.Cin * a.T + co] = acc[n][r];
      else if (j < a.Cin * a.T) {
        const int ci = j / a.T, t = j - ci * a.T;
        prow[((size_t)co * a.Cin + ci) * a.T + t] = acc[n][r];
      }
    }
  }
}

}  // namespace

// Launches the small-Cin kernel when it takes the problem; returns the number of partial rows written
// (the caller reduces them), 0 when the shape stays on the general kernels, < 0 on a launch error.
int pg_wgrad_small_launch(const float* x, const float* dy, float* part, long part_stride, long max_rows,
                          int has_bias, int N, int Cin, int IH, int IW, int Cout, int OH, int OW, int T,
                          const int* tap_dr, const int* tap_dc, int in_act, hipStream_t st) {
  static const bool on = []() { const char* e = PG_AB_ENV("PG_WGRAD_SMALL"); return !(e && e[0] == '0'); }();
  if (!on) return 0;
  // >= 32 output channels: with 16 (ImageGPT's 3x3 input layer) three of the four waves idle and the general
  // kernel measured faster (ImageGPT 100.0 k vs 99.2 k img/s)
  if (Cin > 4 || Cout < 32 || IH != OH || IW != OW || OW % 4 != 0 || Cin * T + (has_bias ? 1 : 0) > SW_COLS || T < 2) return 0;
  if ((((uintptr_t)dy) & 15) != 0) return 0;
  SwArgs a;
  int min_dr = tap_dr[0], max_dr = tap_dr[0], min_dc = tap_dc[0], max_dc = tap_dc[0];
  for (int t = 1; t < T; ++t) {
    min_dr = tap_dr[t] < min_dr ? tap_dr[t] : min_dr; max_dr = tap_dr[t] > max_dr ? tap_dr[t] : max_dr;
    min_dc = tap_dc[t] < min_dc ? tap_dc[t] : min_dc; max_dc = tap_dc[t] > max_dc ? tap_dc[t] : max_dc;
  }
  const int hr = max_dr - min_dr, hc = max_dc - min_dc;
  // rows per tile: about 128 pixels of dy per channel (64 x 132 floats = 33 KB) — two workgroups per CU
  int TR = 128 / OW;
  if (TR < 1) TR = 1;
  if (TR > OH) TR = OH;
  a.x = x; a.dy = dy; a.part = part; a.part_stride = part_stride;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = OH; a.W = OW; a.T = T;
  a.TR = TR; a.tiles_per_img = (OH + TR - 1) / TR; a.total_tiles = N * a.tiles_per_img;
  a.min_dr = min_dr; a.min_dc = min_dc; a.tile_w = OW + hc; a.xrows = TR + hr;
  a.dstride = TR * OW + 4;
  a.in_act = in_act;
  a.ncols = Cin * T;
  a.bias_col = has_bias ? a.ncols : -1;
  for (int j = 0; j < SW_COLS; ++j) a.col_off[j] = -1;
  for (int ci = 0; ci < Cin; ++ci)
    for (int t = 0; t < T; ++t)
      a.col_off[ci * T + t] = ci * a.xrows * a.tile_w + (tap_dr[t] - min_dr) * a.tile_w + (tap_dc[t] - min_dc);
  const size_t shmem = ((size_t)SW_CO * a.dstride + (size_t)Cin * a.xrows * a.tile_w) * sizeof(float);
  if (shmem > 64 * 1024) return 0;
  const int co_chunks = (Cout + SW_CO - 1) / SW_CO;
  long G = 512 / co_chunks;
  if (G > a.total_tiles) G = a.total_tiles;
  if (G > max_rows) G = max_rows;
  if (G < 1) return 0;
  hipLaunchKernelGGL(conv_wgrad_small_kernel, dim3((unsigned)G, (unsigned)co_chunks), dim3(SW_THREADS), shmem, st, a);
  if (hipGetLastError() != hipSuccess) return -1;
  return (int)G;
}
