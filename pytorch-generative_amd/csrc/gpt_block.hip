// gpt_block.hip — everything of an ImageGPT transformer block that is not the attention core, as two
// forward and two backward kernels working on 16-pixel tiles held in registers.
//
// Reference (models/autoregressive/image_gpt.py:21-52, :104-109), C = n_embedding_channels = 16:
//   head:  qkv   = [W_q; W_kv] LN1(x) + [b_q; b_kv]                  (nn/attention.py:134-143)
//   (attention core: attention*.hip)
//   tail:  x_mid = x + W_p o + b_p                                    (x + attn(ln1(x)),   :50)
//          b     = x_mid + W_2 gelu(W_1 LN2(x_mid) + b_1) + b_2       (x + mlp(ln2(x)),    :51-52)
//          x_new = x + b                                              (the model loop adds x again, :107)
// Unfused this is 2 LayerNorms, 5 1x1 convolutions, GELU and 3 adds forward and their 25-odd backward
// launches, each streaming (N, 16..64, L) tensors through HBM. Here a wave owns 16 pixels at a time:
// every [channels x 16 px] intermediate is an MFMA accumulator tile ("D layout": lane (pixel j,
// group g), register r <-> channel 4g + r), and a D-layout tile IS a valid B operand of the next
// v_mfma_f32_16x16x4_f32 (K-step r contracts channels {4g + r}), so the whole chain — projection,
// residual, LayerNorm, fc1, GELU, fc2, residuals — never leaves the register file. Per layer the
// kernels read x, o (resp. d x_new, d qkv) once and write qkv, x_new (resp. d o, d x) once.
// Weight gradients contract over pixels: the computed operand goes through a per-wave LDS transpose,
// the other one is re-read from L2 in the transposed order; partial sums live in registers for all the
// tiles of a wave, are reduced per workgroup in LDS and summed by a small second kernel (deterministic).
// Backward recomputes x_mid, both LayerNorms and the hidden activations instead of storing them.
#include <stdlib.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

constexpr int C = 16, HD = 64, QKV = 48;
constexpr int GB_THREADS = 256;  // 4 waves
constexpr int TS = 20;           // LDS row stride (floats) of a transposed [channel][16 px] tile
constexpr float INV_C = 1.f / 16.f;

// partial-row layouts (floats)
constexpr int T_W1 = 0, T_B1 = T_W1 + HD * C, T_W2 = T_B1 + HD, T_B2 = T_W2 + C * HD, T_WP = T_B2 + C,
              T_BP = T_WP + C * C, T_G2 = T_BP + C, T_BE2 = T_G2 + C, T_PART = T_BE2 + C;          // 2432
constexpr int H_WQ = 0, H_WKV = H_WQ + C * C, H_BQ = H_WKV + 2 * C * C, H_BKV = H_BQ + C,
              H_G1 = H_BKV + 2 * C, H_BE1 = H_G1 + C, H_PART = H_BE1 + C;                            // 848

struct BlockArgs {
  // head
  const float* x; const float* g1; const float* be1; const float* wq; const float* bq;
  const float* wkv; const float* bkv; float* qkv; const float* dqkv; const float* gx; float* dx;
  // tail
  const float* o; const float* wp; const float* bp; const float* g2; const float* be2;
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* xnew; const float* dxnew; float* d_o; float* gx_out;
  float* part;
  int N, L, tiles_per_img, total_tiles;
  float eps;
};

struct Ln { float mu, rs; f32x4 xhat; };

__device__ __forceinline__ float gsum4(float s) {  // over the four lane groups g
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}
__device__ __forceinline__ float jsum16(float s) {  // over the 16 pixel lanes of a group
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
  return s;
}

// LayerNorm statistics of this lane's pixel: its 16 channels are 4 registers x 4 lane groups
// (biased variance, eps inside the square root, one Newton step on v_rsq_f32 — as layernorm.hip)
__device__ __forceinline__ Ln ln_stats(const f32x4& v, float eps) {
  Ln s;
  s.mu = gsum4((v[0] + v[1]) + (v[2] + v[3])) * INV_C;
  const f32x4 dd = {v[0] - s.mu, v[1] - s.mu, v[2] - s.mu, v[3] - s.mu};
  const float a = gsum4(fmaf(dd[0], dd[0], fmaf(dd[1], dd[1], fmaf(dd[2], dd[2], dd[3] * dd[3])))) * INV_C + eps;
  float rs = rsqrtf(a);
  rs = rs * (1.5f - 0.5f * a * rs * rs);
  s.rs = rs;
  s.xhat = f32x4{dd[0] * rs, dd[1] * rs, dd[2] * rs, dd[3] * rs};
  return s;
}

// D-layout tile of a 16-channel tensor: register r <-> channel ch0 + 4g + r of pixel `base`
__device__ __forceinline__ f32x4 load_tile(const float* __restrict__ base, int L, int g) {
  f32x4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = base[(size_t)(4 * g + r) * L];
  return t;
}
__device__ __forceinline__ void store_tile(float* __restrict__ base, int L, int g, const f32x4& t) {
#pragma unroll
  for (int r = 0; r < 4; ++r) base[(size_t)(4 * g + r) * L] = t[r];
}
// per-channel parameter vector in D layout: v[4g + r]
__device__ __forceinline__ f32x4 load_vec(const float* __restrict__ v, int g) {
  return f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
}

__device__ __forceinline__ float gelu_f(float x) { return pg_gelu(x); }

// row `ch` of the merged [W_q; W_kv] matrix (each row has C entries)
__device__ __forceinline__ const float* qkv_row(const BlockArgs& a, int ch) {
  return ch < C ? a.wq + ch * C : a.wkv + (ch - C) * C;
}

// ---------------------------------------------------------------------------------- head, forward
__global__ void __launch_bounds__(GB_THREADS) head_fwd_kernel(const BlockArgs a) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * (GB_THREADS / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (GB_THREADS / 64);
  float wf[3][4];  // A[i = out channel 16m+j][k = g <-> in channel 4g+r]
  f32x4 bias[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      wf[m][r] = qkv_row(a, 16 * m + j)[4 * g + r];
      const int ch = 16 * m + 4 * g + r;
      bias[m][r] = ch < C ? a.bq[ch] : a.bkv[ch - C];
    }
  }
  const f32x4 gam = load_vec(a.g1, g), bet = load_vec(a.be1, g);
  for (int tile = wave; tile < a.total_tiles; tile += nwaves) {
    const int n = tile / a.tiles_per_img;
    const int p = (tile - n * a.tiles_per_img) * 16 + j;
    const f32x4 xv = load_tile(a.x + (size_t)n * C * a.L + p, a.L, g);
    const Ln s = ln_stats(xv, a.eps);
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = fmaf(s.xhat[r], gam[r], bet[r]);
    float* qb = a.qkv + (size_t)n * QKV * a.L + p;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      f32x4 out = bias[m];
#pragma unroll
      for (int r = 0; r < 4; ++r) out = MFMA16(wf[m][r], y[r], out);
      store_tile(qb + (size_t)(16 * m) * a.L, a.L, g, out);
    }
  }
}

// ---------------------------------------------------------------------------------- head, backward
// dx = LN1'(W^T dqkv) + gx ; dW += dqkv^T LN1(x) ; db += sum dqkv ; dgamma1, dbeta1
__global__ void __launch_bounds__(GB_THREADS) head_bwd_kernel(const BlockArgs a) {
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (GB_THREADS / 64) + wv;
  const int nwaves = gridDim.x * (GB_THREADS / 64);
  float* ty = lds + (size_t)wv * C * TS;  // LN1(x)^T [16 c][TS]

  float wt[3][4];  // A[i = in channel j][k = g <-> out channel 16m+4g+r]
#pragma unroll
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) wt[m][r] = qkv_row(a, 16 * m + 4 * g + r)[j];
  }
  const f32x4 gam = load_vec(a.g1, g), bet = load_vec(a.be1, g);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 accw[3], accb[3], dgam = zero4, dbet = zero4;
#pragma unroll
  for (int m = 0; m < 3; ++m) { accw[m] = zero4; accb[m] = zero4; }

  for (int tile = wave; tile < a.total_tiles; tile += nwaves) {
    const int n = tile / a.tiles_per_img;
    const int p0 = (tile - n * a.tiles_per_img) * 16;
    const size_t off = (size_t)n * C * a.L + p0 + j;
    const float* dq = a.dqkv + (size_t)n * QKV * a.L + p0;
    const f32x4 xv = load_tile(a.x + off, a.L, g);
    const f32x4 gxv = load_tile(a.gx + off, a.L, g);
    f32x4 dqv[3], dqt[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      dqv[m] = load_tile(dq + (size_t)(16 * m) * a.L + j, a.L, g);
      // the same tile transposed: channel 16m+j, pixels 4g..4g+3 (A operand of the weight gradient)
      dqt[m] = *reinterpret_cast<const f32x4*>(dq + (size_t)(16 * m + j) * a.L + 4 * g);
    }
    const Ln s = ln_stats(xv, a.eps);
    f32x4 dy = zero4;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dy = MFMA16(wt[m][r], dqv[m][r], dy);
        accb[m][r] += dqv[m][r];
      }
    }
    // LayerNorm backward (dy is the gradient of y = xhat * gamma + beta)
    f32x4 gy;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gy[r] = dy[r] * gam[r];
      s1 += gy[r];
      s2 = fmaf(gy[r], s.xhat[r], s2);
      dgam[r] = fmaf(dy[r], s.xhat[r], dgam[r]);
      dbet[r] += dy[r];
      ty[(4 * g + r) * TS + j] = fmaf(s.xhat[r], gam[r], bet[r]);
    }
    const float m1 = gsum4(s1) * INV_C, m2 = gsum4(s2) * INV_C;
    f32x4 dxv;
#pragma unroll
    for (int r = 0; r < 4; ++r) dxv[r] = s.rs * (gy[r] - m1 - s.xhat[r] * m2) + gxv[r];
    store_tile(a.dx + off, a.L, g, dxv);
    // dW[ch][c] += sum_px dqkv[px][ch] y[px][c]
    const f32x4 yt = *reinterpret_cast<const f32x4*>(ty + j * TS + 4 * g);  // B[k = px 4g+e][c = j]
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int e = 0; e < 4; ++e) accw[m] = MFMA16(dqt[m][e], yt[e], accw[m]);
    }
  }

  __syncthreads();
  float* mine = lds + (size_t)wv * H_PART;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = 16 * m + 4 * g + r;
      mine[H_WQ + ch * C + j] = accw[m][r];  // rows 0..15 = W_q, 16..47 = W_kv: contiguous in the row
      const float b = jsum16(accb[m][r]);
      if (j == 0) mine[H_BQ + ch] = b;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float dg = jsum16(dgam[r]), db = jsum16(dbet[r]);
    if (j == 0) { mine[H_G1 + 4 * g + r] = dg; mine[H_BE1 + 4 * g + r] = db; }
  }
  __syncthreads();
  float* prow = a.part + (size_t)blockIdx.x * H_PART;
  for (int i = threadIdx.x; i < H_PART; i += GB_THREADS)
    prow[i] = (lds[i] + lds[H_PART + i]) + (lds[2 * H_PART + i] + lds[3 * H_PART + i]);
}

// ---------------------------------------------------------------------------------- tail, forward
__global__ void __launch_bounds__(GB_THREADS) tail_fwd_kernel(const BlockArgs a) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * (GB_THREADS / 64) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (GB_THREADS / 64);
  float wpf[4], w1f[4][4], w2f[4][4];
  f32x4 b1r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wpf[r] = a.wp[j * C + 4 * g + r];  // A[i = co j][k <-> ci 4g+r]
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w1f[m][r] = a.w1[(16 * m + j) * C + 4 * g + r];      // A[i = hidden 16m+j][k <-> c 4g+r]
      w2f[m][r] = a.w2[j * HD + 16 * m + 4 * g + r];       // A[i = co j][k <-> hidden 16m+4g+r]
      b1r[m][r] = a.b1[16 * m + 4 * g + r];
    }
  }
  const f32x4 bpv = load_vec(a.bp, g), b2v = load_vec(a.b2, g);
  const f32x4 gam = load_vec(a.g2, g), bet = load_vec(a.be2, g);
  for (int tile = wave; tile < a.total_tiles; tile += nwaves) {
    const int n = tile / a.tiles_per_img;
    const size_t off = (size_t)n * C * a.L + (tile - n * a.tiles_per_img) * 16 + j;
    const f32x4 xv = load_tile(a.x + off, a.L, g);
    const f32x4 ov = load_tile(a.o + off, a.L, g);
    f32x4 xm = xv + bpv;
#pragma unroll
    for (int r = 0; r < 4; ++r) xm = MFMA16(wpf[r], ov[r], xm);
    const Ln s = ln_stats(xm, a.eps);
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = fmaf(s.xhat[r], gam[r], bet[r]);
    f32x4 h[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      h[m] = b1r[m];
#pragma unroll
      for (int r = 0; r < 4; ++r) h[m] = MFMA16(w1f[m][r], y[r], h[m]);
    }
    f32x4 out = (xm + b2v) + xv;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) out = MFMA16(w2f[m][r], gelu_f(h[m][r]), out);
    }
    store_tile(a.xnew + off, a.L, g, out);
  }
}

// ---------------------------------------------------------------------------------- tail, backward
// D = d x_new. Outputs d_o = W_p^T d x_mid and gx = D + d x_mid (everything that reaches the block
// input x except through LN1), plus the gradients of W_p, b_p, LN2, fc1, fc2.
__global__ void __launch_bounds__(GB_THREADS) tail_bwd_kernel(const BlockArgs a) {
  extern __shared__ float4 lds4[];
  float* lds = reinterpret_cast<float*>(lds4);
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * (GB_THREADS / 64) + wv;
  const int nwaves = gridDim.x * (GB_THREADS / 64);
  float* tg = lds + (size_t)wv * (2 * HD + 2 * C) * TS;  // G^T     [64][TS]
  float* th = tg + HD * TS;                              // dH^T    [64][TS]
  float* ty = th + HD * TS;                              // LN2^T   [16][TS]
  float* tx = ty + C * TS;                               // dx_mid^T[16][TS]

  // Weight fragments depend on the lane only, not on the wave or the tile: the workgroup keeps ONE
  // copy in LDS ([fragment][lane], conflict-free ds_read_b32) instead of 64 VGPRs per lane — with
  // them in registers the kernel needs 316 registers (one wave per SIMD, measured 196 us per launch).
  //   w1f[m][r] = W1[16m+j][4g+r]      A[i = hidden 16m+j][k <-> c 4g+r]       (H recompute)
  //   w2t[m][r] = W2[4g+r][16m+j]      A[i = hidden 16m+j][k <-> co 4g+r]      (dG = W_2^T D)
  //   w1t[m][r] = W1[16m+4g+r][j]      A[i = c j][k <-> hidden 16m+4g+r]       (dY = W_1^T dH)
  //   b1r[m][r] = b1[16m+4g+r]
  float* wl = lds + (size_t)4 * (2 * HD + 2 * C) * TS;  // [64 fragments][64 lanes]
  if (wv == 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        wl[(0 + 4 * m + r) * 64 + lane] = a.w1[(16 * m + j) * C + 4 * g + r];
        wl[(16 + 4 * m + r) * 64 + lane] = a.w2[(4 * g + r) * HD + 16 * m + j];
        wl[(32 + 4 * m + r) * 64 + lane] = a.w1[(16 * m + 4 * g + r) * C + j];
        wl[(48 + 4 * m + r) * 64 + lane] = a.b1[16 * m + 4 * g + r];
      }
    }
  }
  __syncthreads();
  const float* wlane = wl + lane;
#define W1F(m, r) wlane[(0 + 4 * (m) + (r)) * 64]
#define W2T(m, r) wlane[(16 + 4 * (m) + (r)) * 64]
#define W1T(m, r) wlane[(32 + 4 * (m) + (r)) * 64]
#define B1R(m, r) wlane[(48 + 4 * (m) + (r)) * 64]
  float wpf[4], wpt[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    wpf[r] = a.wp[j * C + 4 * g + r];    // A[i = co j][k <-> ci 4g+r]      (x_mid recompute)
    wpt[r] = a.wp[(4 * g + r) * C + j];  // A[i = ci j][k <-> co 4g+r]      (d_o = W_p^T d x_mid)
  }
  const f32x4 bpv = load_vec(a.bp, g);
  const f32x4 gam = load_vec(a.g2, g), bet = load_vec(a.be2, g);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc1[4], acc2[4], db1[4], accp = zero4, db2 = zero4, dbp = zero4, dgam = zero4, dbet = zero4;
#pragma unroll
  for (int m = 0; m < 4; ++m) { acc1[m] = zero4; acc2[m] = zero4; db1[m] = zero4; }

  for (int tile = wave; tile < a.total_tiles; tile += nwaves) {
    const int n = tile / a.tiles_per_img;
    const int p0 = (tile - n * a.tiles_per_img) * 16;
    const size_t img = (size_t)n * C * a.L + p0;
    const size_t off = img + j;
    const f32x4 xv = load_tile(a.x + off, a.L, g);
    const f32x4 ov = load_tile(a.o + off, a.L, g);
    const f32x4 dv = load_tile(a.dxnew + off, a.L, g);
    // transposed fragments (channel j, pixels 4g..4g+3) for the weight gradients: L2 hits
    const f32x4 dvt = *reinterpret_cast<const f32x4*>(a.dxnew + img + (size_t)j * a.L + 4 * g);
    const f32x4 ot = *reinterpret_cast<const f32x4*>(a.o + img + (size_t)j * a.L + 4 * g);

    // ---- recompute the forward: x_mid, LN2, hidden
    f32x4 xm = xv + bpv;
#pragma unroll
    for (int r = 0; r < 4; ++r) xm = MFMA16(wpf[r], ov[r], xm);
    const Ln s = ln_stats(xm, a.eps);
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      y[r] = fmaf(s.xhat[r], gam[r], bet[r]);
      ty[(4 * g + r) * TS + j] = y[r];
      db2[r] += dv[r];
    }
    f32x4 h[4], dg[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      h[m] = f32x4{B1R(m, 0), B1R(m, 1), B1R(m, 2), B1R(m, 3)};
      dg[m] = zero4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h[m] = MFMA16(W1F(m, r), y[r], h[m]);
        dg[m] = MFMA16(W2T(m, r), dv[r], dg[m]);
      }
    }
    // ---- G, dH = dG * gelu'(H); dY = W_1^T dH
    f32x4 dy = zero4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hv = h[m][r];
        float cdf, ee;
        pg_gelu_parts(hv, cdf, ee);
        const float pdf = 0.39894228040143267794f * ee;
        const float dh = dg[m][r] * (cdf + hv * pdf);
        const int hid = 16 * m + 4 * g + r;
        tg[hid * TS + j] = hv * cdf;
        th[hid * TS + j] = dh;
        db1[m][r] += dh;
        dy = MFMA16(W1T(m, r), dh, dy);
      }
    }
    // ---- LN2 backward, d x_mid = D + LN2'(dY)
    f32x4 gy, dxm;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gy[r] = dy[r] * gam[r];
      s1 += gy[r];
      s2 = fmaf(gy[r], s.xhat[r], s2);
      dgam[r] = fmaf(dy[r], s.xhat[r], dgam[r]);
      dbet[r] += dy[r];
    }
    const float m1 = gsum4(s1) * INV_C, m2 = gsum4(s2) * INV_C;
    f32x4 dov = zero4, gxv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dxm[r] = dv[r] + s.rs * (gy[r] - m1 - s.xhat[r] * m2);
      tx[(4 * g + r) * TS + j] = dxm[r];
      dbp[r] += dxm[r];
      gxv[r] = dv[r] + dxm[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dov = MFMA16(wpt[r], dxm[r], dov);
    store_tile(a.d_o + off, a.L, g, dov);
    store_tile(a.gx_out + off, a.L, g, gxv);

    // ---- weight gradients: contraction over the tile's pixels (pixel 4g+e in K-step e)
    const f32x4 yt = *reinterpret_cast<const f32x4*>(ty + j * TS + 4 * g);   // B[k = px][c = j]
    const f32x4 xt = *reinterpret_cast<const f32x4*>(tx + j * TS + 4 * g);   // A[i = co j][k = px]
#pragma unroll
    for (int e = 0; e < 4; ++e) accp = MFMA16(xt[e], ot[e], accp);            // dWp[co][ci]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 gt = *reinterpret_cast<const f32x4*>(tg + (16 * m + j) * TS + 4 * g);  // B[k = px][hidden]
      const f32x4 ht = *reinterpret_cast<const f32x4*>(th + (16 * m + j) * TS + 4 * g);  // A[hidden][k = px]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc2[m] = MFMA16(dvt[e], gt[e], acc2[m]);  // dW2[co][hidden] += D[px][co] G[px][hidden]
        acc1[m] = MFMA16(ht[e], yt[e], acc1[m]);   // dW1[hidden][c]  += dH[px][hidden] y[px][c]
      }
    }
  }

#undef W1F
#undef W2T
#undef W1T
#undef B1R
  // ---- one partial row per workgroup
  __syncthreads();
  float* mine = lds + (size_t)wv * T_PART;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mine[T_W1 + (16 * m + 4 * g + r) * C + j] = acc1[m][r];    // D: lane c = j, rows hidden 16m+4g+r
      mine[T_W2 + (4 * g + r) * HD + 16 * m + j] = acc2[m][r];   // D: lane hidden 16m+j, rows co 4g+r
      const float v = jsum16(db1[m][r]);
      if (j == 0) mine[T_B1 + 16 * m + 4 * g + r] = v;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    mine[T_WP + (4 * g + r) * C + j] = accp[r];                  // D: lane ci = j, rows co 4g+r
    const float v2 = jsum16(db2[r]), vp = jsum16(dbp[r]), vg = jsum16(dgam[r]), vb = jsum16(dbet[r]);
    if (j == 0) {
      mine[T_B2 + 4 * g + r] = v2;
      mine[T_BP + 4 * g + r] = vp;
      mine[T_G2 + 4 * g + r] = vg;
      mine[T_BE2 + 4 * g + r] = vb;
    }
  }
  __syncthreads();
  float* prow = a.part + (size_t)blockIdx.x * T_PART;
  for (int i = threadIdx.x; i < T_PART; i += GB_THREADS)
    prow[i] = (lds[i] + lds[T_PART + i]) + (lds[2 * T_PART + i] + lds[3 * T_PART + i]);
}

// ---- second stage: out_k[i] += sum_rows part[row][seg_k + i] for up to 8 gradient tensors
struct SegArgs {
  const float* part; int rows, stride, nseg;
  int end[8]; float* dst[8];
};
__global__ void __launch_bounds__(256) seg_reduce_kernel(const SegArgs a) {
  __shared__ float red[32][9];
  const int sl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int s = blockIdx.x * 8 + sl;
  float a0 = 0.f, a1 = 0.f;
  if (s < a.stride) {
    const float* p = a.part + s;
    int r = rg;
    for (; r + 32 < a.rows; r += 64) {
      a0 += p[(size_t)r * a.stride];
      a1 += p[(size_t)(r + 32) * a.stride];
    }
    if (r < a.rows) a0 += p[(size_t)r * a.stride];
  }
  red[rg][sl] = a0 + a1;
  __syncthreads();
  if (rg != 0 || s >= a.stride) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) acc += red[r][sl];
  int begin = 0;
  for (int k = 0; k < a.nseg; ++k) {
    if (s < a.end[k]) { a.dst[k][s - begin] += acc; return; }
    begin = a.end[k];
  }
}

// Two segmented reductions in ONE launch (round 3): the tail and head kernels of a block each leave rows of
// partial weight-gradient sums; at the reference's batch 64 the 16 reduce launches of a step were 6 % of it.
struct SegArgs2 { SegArgs j[2]; int blocks0; };
__global__ void __launch_bounds__(256) seg_reduce2_kernel(const SegArgs2 a2) {
  __shared__ float red[32][9];
  const bool second = (int)blockIdx.x >= a2.blocks0;
  const SegArgs& a = a2.j[second ? 1 : 0];
  const int blk = second ? blockIdx.x - a2.blocks0 : blockIdx.x;
  const int sl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int s = blk * 8 + sl;
  float a0 = 0.f, a1 = 0.f;
  if (s < a.stride) {
    const float* p = a.part + s;
    int r = rg;
    for (; r + 32 < a.rows; r += 64) {
      a0 += p[(size_t)r * a.stride];
      a1 += p[(size_t)(r + 32) * a.stride];
    }
    if (r < a.rows) a0 += p[(size_t)r * a.stride];
  }
  red[rg][sl] = a0 + a1;
  __syncthreads();
  if (rg != 0 || s >= a.stride) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) acc += red[r][sl];
  int begin = 0;
  for (int k = 0; k < a.nseg; ++k) {
    if (s < a.end[k]) { a.dst[k][s - begin] += acc; return; }
    begin = a.end[k];
  }
}

// Round 6: the reductions of up to 8 blocks (16 jobs) in ONE launch at the end of the backward pass — at the reference's batch 64 a
// replayed step is ~83 launches of which ~45 run at the ~4.5 us floor of a graph kernel node; the eight per-block reduce launches
// were 3.9 % of its kernel time (profiles/r05_image_gpt_b64_kernel_stats.csv).
constexpr int SEG_MAX_JOBS = 16;
struct SegArgsN { SegArgs j[SEG_MAX_JOBS]; int first[SEG_MAX_JOBS + 1]; int n; };
__global__ void __launch_bounds__(256) seg_reduceN_kernel(const SegArgsN an) {
  __shared__ float red[32][9];
  int job = 0;
  while (job + 1 < an.n && (int)blockIdx.x >= an.first[job + 1]) ++job;
  const SegArgs& a = an.j[job];
  const int blk = blockIdx.x - an.first[job];
  const int sl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int s = blk * 8 + sl;
  float a0 = 0.f, a1 = 0.f;
  if (s < a.stride) {
    const float* p = a.part + s;
    int r = rg;
    for (; r + 32 < a.rows; r += 64) {
      a0 += p[(size_t)r * a.stride];
      a1 += p[(size_t)(r + 32) * a.stride];
    }
    if (r < a.rows) a0 += p[(size_t)r * a.stride];
  }
  red[rg][sl] = a0 + a1;
  __syncthreads();
  if (rg != 0 || s >= a.stride) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) acc += red[r][sl];
  int begin = 0;
  for (int k = 0; k < a.nseg; ++k) {
    if (s < a.end[k]) { a.dst[k][s - begin] += acc; return; }
    begin = a.end[k];
  }
}

// Workgroups per launch: whole rounds of resident waves (256 CUs x 4 SIMDs; the kernels hold 5 / 3 / 4 / 2
// waves per SIMD by their register counts), at least `min_tiles` tiles per wave. which: 0 head fwd,
// 1 head bwd, 2 tail fwd, 3 tail bwd. PG_BLOCK_GRID="a,b,c,d" overrides the caps (tuning).
int grid_blocks(int which, int N, int L) {
  // immutable init-once tables (function-local static with an initialiser: thread-safe; the library
  // is entered from the main thread and from the autograd thread)
  struct Cfg { int cap[4]; int mt[2]; };
  static const Cfg cfg = []() {
    Cfg c = {{2048, 1024, 2048, 512},  // measured: head bwd 79.8 us at 512, 68.9 at 768; round 5 (122 registers = four waves per SIMD since the
                                      // VGPR-form build): 1024 = one full round, ImageGPT 99.16 -> 99.88 k img/s on one box (sweep in profiles/README.md)
             {1, 2}};                 // tiles per wave below which the grid shrinks (forward, backward);
                                      // measured at batch 64: (4, 8) 1.82 ms/step, (2, 4) 1.59, (1, 2) 1.53, (1, 1) 1.56
    if (const char* e = PG_AB_ENV("PG_BLOCK_GRID")) {
      int v[4];
      if (sscanf(e, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4 && v[0] > 0 && v[1] > 0 && v[2] > 0 && v[3] > 0)
        for (int i = 0; i < 4; ++i) c.cap[i] = v[i];
    }
    if (const char* e = PG_AB_ENV("PG_BLOCK_MINTILES")) {
      int f = 0, b = 0;
      if (sscanf(e, "%d,%d", &f, &b) == 2 && f > 0 && b > 0) { c.mt[0] = f; c.mt[1] = b; }
    }
    return c;
  }();
  const int* cap = cfg.cap;
  const int* mt_cfg = cfg.mt;
  const int min_tiles = (which & 1) ? mt_cfg[1] : mt_cfg[0];
  const long tiles = (long)N * (L / 16);
  long b = (tiles + 4 * min_tiles - 1) / (4 * min_tiles);
  if (b > cap[which]) b = cap[which];
  return b < 1 ? 1 : (int)b;
}
int bwd_blocks(int N, int L) { return grid_blocks(3, N, L); }

int check_shape(const char* who, int N, int Cc, int L) {
  PG_REQUIRE(N > 0 && L > 0, PG_EINVAL, "%s: non-positive dimension", who);
  PG_REQUIRE(Cc == C, PG_ESHAPE, "%s: only C = 16 is instantiated (got %d)", who, Cc);
  PG_REQUIRE(L % 16 == 0, PG_ESHAPE, "%s: L=%d is not a multiple of 16", who, L);
  return 0;
}

void set_geometry(BlockArgs& a, int N, int L, float eps) {
  a.N = N; a.L = L; a.tiles_per_img = L / 16; a.total_tiles = N * a.tiles_per_img; a.eps = eps;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

PG_EXPORT size_t pg_gpt_block_head_bwd_workspace_floats(int N, int L) {
  if (N <= 0 || L < 16) return 0;
  return (size_t)grid_blocks(1, N, L) * H_PART;
}
PG_EXPORT size_t pg_gpt_block_tail_bwd_workspace_floats(int N, int L) {
  if (N <= 0 || L < 16) return 0;
  return (size_t)grid_blocks(3, N, L) * T_PART;
}

PG_EXPORT int pg_gpt_block_head_fwd(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                                    const float* bq, const float* wkv, const float* bkv, float* qkv,
                                    int N, int Cc, int L, float eps, void* stream) {
  PG_REQUIRE(x && ln_w && ln_b && wq && bq && wkv && bkv && qkv, PG_EINVAL, "pg_gpt_block_head_fwd: null pointer");
  int rc = check_shape("pg_gpt_block_head_fwd", N, Cc, L);
  if (rc) return rc;
  BlockArgs a = {};
  a.x = x; a.g1 = ln_w; a.be1 = ln_b; a.wq = wq; a.bq = bq; a.wkv = wkv; a.bkv = bkv; a.qkv = qkv;
  set_geometry(a, N, L, eps);
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)grid_blocks(0, N, L)), dim3(GB_THREADS), 0, (hipStream_t)stream, a);
  PG_LAUNCH_CHECK("pg_gpt_block_head_fwd");
  return 0;
}

namespace {
SegArgs tail_seg_args(const float* workspace, int rows, float* dw1, float* db1, float* dw2, float* db2,
                      float* dwp, float* dbp, float* dln_w, float* dln_b) {
  SegArgs r = {};
  r.part = workspace; r.rows = rows; r.stride = T_PART; r.nseg = 8;
  r.end[0] = T_B1; r.dst[0] = dw1;
  r.end[1] = T_W2; r.dst[1] = db1;
  r.end[2] = T_B2; r.dst[2] = dw2;
  r.end[3] = T_WP; r.dst[3] = db2;
  r.end[4] = T_BP; r.dst[4] = dwp;
  r.end[5] = T_G2; r.dst[5] = dbp;
  r.end[6] = T_BE2; r.dst[6] = dln_w;
  r.end[7] = T_PART; r.dst[7] = dln_b;
  return r;
}

SegArgs head_seg_args(const float* workspace, int rows, float* dln_w, float* dln_b, float* dwq, float* dbq, float* dwkv,
                      float* dbkv) {
  SegArgs r = {};
  r.part = workspace; r.rows = rows; r.stride = H_PART; r.nseg = 6;
  r.end[0] = H_WKV; r.dst[0] = dwq;
  r.end[1] = H_BQ; r.dst[1] = dwkv;
  r.end[2] = H_BKV; r.dst[2] = dbq;
  r.end[3] = H_G1; r.dst[3] = dbkv;
  r.end[4] = H_BE1; r.dst[4] = dln_w;
  r.end[5] = H_PART; r.dst[5] = dln_b;
  return r;
}

// defer != 0: the partial rows stay in `workspace` (pg_gpt_blocks_reduce adds them later); the gradient pointers are then unused
int head_bwd_impl(const float* x, const float* ln_w, const float* ln_b, const float* wq, const float* wkv,
                  const float* dqkv, const float* gx, float* dx, float* dln_w, float* dln_b, float* dwq,
                  float* dbq, float* dwkv, float* dbkv, int N, int Cc, int L, float eps, float* workspace,
                  size_t workspace_floats, const SegArgs* tail, void* stream, int defer = 0) {
  PG_REQUIRE(x && ln_w && ln_b && wq && wkv && dqkv && gx && dx && workspace &&
                 (defer || (dln_w && dln_b && dwq && dbq && dwkv && dbkv)), PG_EINVAL, "pg_gpt_block_head_bwd: null pointer");
  int rc = check_shape("pg_gpt_block_head_bwd", N, Cc, L);
  if (rc) return rc;
  PG_REQUIRE(workspace_floats >= pg_gpt_block_head_bwd_workspace_floats(N, L), PG_EINVAL,
             "pg_gpt_block_head_bwd: workspace too small");
  PG_REQUIRE(al16(dqkv), PG_EINVAL, "pg_gpt_block_head_bwd: dqkv must be 16-byte aligned");
  BlockArgs a = {};
  a.x = x; a.g1 = ln_w; a.be1 = ln_b; a.wq = wq; a.wkv = wkv; a.dqkv = dqkv; a.gx = gx; a.dx = dx;
  a.part = workspace;
  set_geometry(a, N, L, eps);
  const int blocks = grid_blocks(1, N, L);
  hipStream_t st = (hipStream_t)stream;
  const size_t tr = (size_t)4 * C * TS, rd = (size_t)4 * H_PART;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)blocks), dim3(GB_THREADS), (tr > rd ? tr : rd) * sizeof(float), st, a);
  PG_LAUNCH_CHECK("pg_gpt_block_head_bwd");
  if (defer) return 0;
  const SegArgs r = head_seg_args(workspace, blocks, dln_w, dln_b, dwq, dbq, dwkv, dbkv);
  if (tail) {  // one launch for this block's two reductions
    SegArgs2 r2;
    r2.j[0] = r; r2.j[1] = *tail; r2.blocks0 = (H_PART + 7) / 8;
    hipLaunchKernelGGL(seg_reduce2_kernel, dim3((unsigned)(r2.blocks0 + (T_PART + 7) / 8)), dim3(256), 0, st, r2);
  } else {
    hipLaunchKernelGGL(seg_reduce_kernel, dim3((unsigned)((H_PART + 7) / 8)), dim3(256), 0, st, r);
  }
  PG_LAUNCH_CHECK("pg_gpt_block_head_bwd(reduce)");
  return 0;
}
}  // namespace

PG_EXPORT int pg_gpt_block_head_bwd(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                                    const float* wkv, const float* dqkv, const float* gx, float* dx,
                                    float* dln_w, float* dln_b, float* dwq, float* dbq, float* dwkv,
                                    float* dbkv, int N, int Cc, int L, float eps, float* workspace,
                                    size_t workspace_floats, void* stream) {
  return head_bwd_impl(x, ln_w, ln_b, wq, wkv, dqkv, gx, dx, dln_w, dln_b, dwq, dbq, dwkv, dbkv, N, Cc, L, eps,
                       workspace, workspace_floats, nullptr, stream);
}

// head backward + the reduction a preceding pg_gpt_block_tail_bwd_partial of the SAME block left undone
PG_EXPORT int pg_gpt_block_head_bwd_with_tail(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                                              const float* wkv, const float* dqkv, const float* gx, float* dx,
                                              float* dln_w, float* dln_b, float* dwq, float* dbq, float* dwkv,
                                              float* dbkv, int N, int Cc, int L, float eps, float* workspace,
                                              size_t workspace_floats, const float* tail_workspace, float* t_dw1,
                                              float* t_db1, float* t_dw2, float* t_db2, float* t_dwp, float* t_dbp,
                                              float* t_dln_w, float* t_dln_b, void* stream) {
  PG_REQUIRE(tail_workspace && t_dw1 && t_db1 && t_dw2 && t_db2 && t_dwp && t_dbp && t_dln_w && t_dln_b, PG_EINVAL,
             "pg_gpt_block_head_bwd_with_tail: null pointer");
  if (int rc = check_shape("pg_gpt_block_head_bwd_with_tail", N, Cc, L)) return rc;
  const SegArgs tail = tail_seg_args(tail_workspace, bwd_blocks(N, L), t_dw1, t_db1, t_dw2, t_db2, t_dwp, t_dbp,
                                     t_dln_w, t_dln_b);
  return head_bwd_impl(x, ln_w, ln_b, wq, wkv, dqkv, gx, dx, dln_w, dln_b, dwq, dbq, dwkv, dbkv, N, Cc, L, eps,
                       workspace, workspace_floats, &tail, stream);
}

// head backward WITHOUT its reduction: the partial rows stay in `workspace` for pg_gpt_blocks_reduce
PG_EXPORT int pg_gpt_block_head_bwd_partial(const float* x, const float* ln_w, const float* ln_b, const float* wq,
                                            const float* wkv, const float* dqkv, const float* gx, float* dx, int N, int Cc,
                                            int L, float eps, float* workspace, size_t workspace_floats, void* stream) {
  return head_bwd_impl(x, ln_w, ln_b, wq, wkv, dqkv, gx, dx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, Cc, L,
                       eps, workspace, workspace_floats, nullptr, stream, 1);
}

// ONE launch for the weight-gradient reductions of n_blocks (<= 8) blocks whose head (pg_gpt_block_head_bwd_partial) and tail
// (pg_gpt_block_tail_bwd_partial) kernels left their partial rows. grads: n_blocks x 14 destinations, per block in the order
// dln1_w, dln1_b, dwq, dbq, dwkv, dbkv (head) | t_dw1, t_db1, t_dw2, t_db2, t_dwp, t_dbp, t_dln_w, t_dln_b (tail); added to.
PG_EXPORT int pg_gpt_blocks_reduce(int n_blocks, const float* const* head_ws, const float* const* tail_ws, float* const* grads,
                                   int N, int Cc, int L, void* stream) {
  PG_REQUIRE(head_ws && tail_ws && grads, PG_EINVAL, "pg_gpt_blocks_reduce: null pointer");
  PG_REQUIRE(n_blocks >= 1 && 2 * n_blocks <= SEG_MAX_JOBS, PG_ESHAPE, "pg_gpt_blocks_reduce: 1..8 blocks per launch, got %d", n_blocks);
  if (int rc = check_shape("pg_gpt_blocks_reduce", N, Cc, L)) return rc;
  SegArgsN an = {};
  int nb = 0;
  for (int b = 0; b < n_blocks; ++b) {
    float* const* g = grads + 14 * b;
    PG_REQUIRE(head_ws[b] && tail_ws[b], PG_EINVAL, "pg_gpt_blocks_reduce: null workspace of block %d", b);
    for (int k = 0; k < 14; ++k) PG_REQUIRE(g[k], PG_EINVAL, "pg_gpt_blocks_reduce: null gradient %d of block %d", k, b);
    an.j[2 * b] = head_seg_args(head_ws[b], grid_blocks(1, N, L), g[0], g[1], g[2], g[3], g[4], g[5]);
    an.first[2 * b] = nb; nb += (H_PART + 7) / 8;
    an.j[2 * b + 1] = tail_seg_args(tail_ws[b], bwd_blocks(N, L), g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13]);
    an.first[2 * b + 1] = nb; nb += (T_PART + 7) / 8;
  }
  an.n = 2 * n_blocks;
  an.first[an.n] = nb;
  hipLaunchKernelGGL(seg_reduceN_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, an);
  PG_LAUNCH_CHECK("pg_gpt_blocks_reduce");
  return 0;
}

PG_EXPORT int pg_gpt_block_tail_fwd(const float* o, const float* x, const float* wp, const float* bp,
                                    const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* x_new, int N, int Cc, int Hd,
                                    int L, float eps, void* stream) {
  PG_REQUIRE(o && x && wp && bp && ln_w && ln_b && w1 && b1 && w2 && b2 && x_new, PG_EINVAL,
             "pg_gpt_block_tail_fwd: null pointer");
  int rc = check_shape("pg_gpt_block_tail_fwd", N, Cc, L);
  if (rc) return rc;
  PG_REQUIRE(Hd == HD, PG_ESHAPE, "pg_gpt_block_tail_fwd: only hidden = 64 is instantiated (got %d)", Hd);
  BlockArgs a = {};
  a.o = o; a.x = x; a.wp = wp; a.bp = bp; a.g2 = ln_w; a.be2 = ln_b; a.w1 = w1; a.b1 = b1; a.w2 = w2;
  a.b2 = b2; a.xnew = x_new;
  set_geometry(a, N, L, eps);
  hipLaunchKernelGGL(tail_fwd_kernel, dim3((unsigned)grid_blocks(2, N, L)), dim3(GB_THREADS), 0, (hipStream_t)stream, a);
  PG_LAUNCH_CHECK("pg_gpt_block_tail_fwd");
  return 0;
}

namespace {
int tail_bwd_impl(const float* o, const float* x, const float* wp, const float* bp,
                                    const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                                    const float* w2, const float* dx_new, float* d_o, float* gx,
                                    float* dwp, float* dbp, float* dln_w, float* dln_b, float* dw1,
                                    float* db1, float* dw2, float* db2, int N, int Cc, int Hd, int L,
                                    float eps, float* workspace, size_t workspace_floats, bool reduce_now, void* stream) {
  PG_REQUIRE(o && x && wp && bp && ln_w && ln_b && w1 && b1 && w2 && dx_new && d_o && gx && dwp && dbp &&
                 dln_w && dln_b && dw1 && db1 && dw2 && db2 && workspace, PG_EINVAL,
             "pg_gpt_block_tail_bwd: null pointer");
  int rc = check_shape("pg_gpt_block_tail_bwd", N, Cc, L);
  if (rc) return rc;
  PG_REQUIRE(Hd == HD, PG_ESHAPE, "pg_gpt_block_tail_bwd: only hidden = 64 is instantiated (got %d)", Hd);
  PG_REQUIRE(workspace_floats >= pg_gpt_block_tail_bwd_workspace_floats(N, L), PG_EINVAL,
             "pg_gpt_block_tail_bwd: workspace too small");
  PG_REQUIRE(al16(o) && al16(dx_new), PG_EINVAL, "pg_gpt_block_tail_bwd: o / dx_new must be 16-byte aligned");
  BlockArgs a = {};
  a.o = o; a.x = x; a.wp = wp; a.bp = bp; a.g2 = ln_w; a.be2 = ln_b; a.w1 = w1; a.b1 = b1; a.w2 = w2;
  a.dxnew = dx_new; a.d_o = d_o; a.gx_out = gx; a.part = workspace;
  set_geometry(a, N, L, eps);
  const int blocks = bwd_blocks(N, L);
  hipStream_t st = (hipStream_t)stream;
  const size_t tr = (size_t)4 * (2 * HD + 2 * C) * TS + 64 * 64, rd = (size_t)4 * T_PART;
  const size_t shmem = (tr > rd ? tr : rd) * sizeof(float);  // 66.4 KB: above the 64 KB default
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(tail_bwd_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  (void)attr;  // thread-safe one-time opt-in
  hipLaunchKernelGGL(tail_bwd_kernel, dim3((unsigned)blocks), dim3(GB_THREADS), shmem, st, a);
  PG_LAUNCH_CHECK("pg_gpt_block_tail_bwd");
  if (!reduce_now) return 0;  // the partial rows stay in `workspace` for pg_gpt_block_head_bwd_with_tail
  const SegArgs r = tail_seg_args(workspace, blocks, dw1, db1, dw2, db2, dwp, dbp, dln_w, dln_b);
  hipLaunchKernelGGL(seg_reduce_kernel, dim3((unsigned)((T_PART + 7) / 8)), dim3(256), 0, st, r);
  PG_LAUNCH_CHECK("pg_gpt_block_tail_bwd(reduce)");
  return 0;
}
}  // namespace

PG_EXPORT int pg_gpt_block_tail_bwd(const float* o, const float* x, const float* wp, const float* bp,
                                    const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                                    const float* w2, const float* dx_new, float* d_o, float* gx,
                                    float* dwp, float* dbp, float* dln_w, float* dln_b, float* dw1,
                                    float* db1, float* dw2, float* db2, int N, int Cc, int Hd, int L,
                                    float eps, float* workspace, size_t workspace_floats, void* stream) {
  return tail_bwd_impl(o, x, wp, bp, ln_w, ln_b, w1, b1, w2, dx_new, d_o, gx, dwp, dbp, dln_w, dln_b, dw1, db1, dw2,
                       db2, N, Cc, Hd, L, eps, workspace, workspace_floats, true, stream);
}

// the tail kernel only: its rows of partial weight-gradient sums stay in `workspace` (which must stay alive) until
// pg_gpt_block_head_bwd_with_tail of the same block reduces them together with its own
PG_EXPORT int pg_gpt_block_tail_bwd_partial(const float* o, const float* x, const float* wp, const float* bp,
                                            const float* ln_w, const float* ln_b, const float* w1, const float* b1,
                                            const float* w2, const float* dx_new, float* d_o, float* gx, int N,
                                            int Cc, int Hd, int L, float eps, float* workspace,
                                            size_t workspace_floats, void* stream) {
  float* dummy = workspace;  // the gradient destinations are not touched without the reduction
  return tail_bwd_impl(o, x, wp, bp, ln_w, ln_b, w1, b1, w2, dx_new, d_o, gx, dummy, dummy, dummy, dummy, dummy, dummy,
                       dummy, dummy, N, Cc, Hd, L, eps, workspace, workspace_floats, false, stream);
}
