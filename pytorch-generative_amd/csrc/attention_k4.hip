// attention_k4.hip — causal attention core for d_k = 4*DKT, d_v = 16*DVT on the fp32 matrix cores,
// gfx950: PixelSNAIL (1 head, d_k 4, d_v 32, strict mask, L = 1024) and, since round 3, ImageGPT at the
// reference's reproduce() shape (image_gpt.py:147-154: 64 embedding channels / 2 heads -> d_k = d_v =
// 32, L = 784). For d_k >= 16 the score tile is a chain of DKT MFMAs over the channel quads and dQ / dK
// are MFMA accumulations like P.V / dV; for d_k = 4 they stay on the VALU (M = 4 would waste 75 % of a
// tile). L must be a multiple of 16; a wave's last 16-row groups may lie beyond L (skipped).
//
// Replaces the body of CausalAttention.forward after the projections (reference
// nn/attention.py:147-160: q k^T / sqrt(d_k), masked_fill, softmax, masked_fill(0), @ v) as
// instantiated at models/autoregressive/pixel_snail.py:92-98, and its autograd.
//
// d_k = 4 is exactly the K extent of v_mfma_f32_16x16x4_f32, so a 16-key x 16-query score tile is
// ONE MFMA, and with d_v = 32 the P.V / dP / dV products are full 16x16x4 tiles with no padding:
//   S^T[key][query]   = K[16 x 4] . Q^T[4 x 16]                     1 MFMA   (C operand = -max / -lse2)
//   O^T[dv][query]   += V^T[16 dv x 4 keys] . P^T[4 keys x 16]      2 DVT MFMAs x 4 key quads
// The score accumulator holds, in lane (g = lane>>4, j = lane&15), keys 4g..4g+3 of query j; its
// register r is directly the B operand of the r-th P.V MFMA (K index g <-> key 4g + r) while the A
// operand is a float4 of a V^T row — no cross-lane movement anywhere in the main loops.
//
// CDNA4 mapping: no LDS and no workgroup cooperation. A wave (= workgroup of 64 lanes) owns the
// 64- (or 32-) query block b of one (image, head) and then block NB-1-b, so every wave walks the same number
// of key tiles of the causal triangle; K / V / dO fragments are read straight from the NCHW planes
// (64-byte runs per channel row) and come from L2: all waves of an image are mapped to ONE XCD
// (blockIdx % 8 selects the XCD), so an image's K/V enter that XCD's L2 once. Forward is two-pass
// (row max with the score MFMA only, then exp / accumulate against the final max: no rescaling).
#include "attention_args.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA4(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

constexpr float NEG_BIG = -1.0e30f;
constexpr float LSE_EMPTY = 1.0e30f;  // lse2 of a query with no allowed key: exp2(s - lse2) == 0

struct K4Map {
  int unit, b0, b1;  // (image, head) unit; the two 64-row blocks of this wave (b1 < 0: none)
};

// blockIdx -> (unit, block pair). Units are dealt round-robin to the 8 XCDs (blockIdx % 8 is the
// XCD), 8 units per XCD at a time, heaviest pair first inside a group of units.
__device__ __forceinline__ K4Map k4_map(int units, int NB) {
  const int npair = (NB + 1) >> 1;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int G = 8;
  const int grp = slot / (G * npair);
  const int rem = slot - grp * (G * npair);
  const int pr = rem / G;
  const int ug = rem - pr * G;
  K4Map m;
  m.unit = (grp * G + ug) * 8 + xcd;
  m.b0 = NB - 1 - pr;
  m.b1 = (pr < NB - 1 - pr) ? pr : -1;
  if (m.unit >= units) m.unit = -1;
  return m;
}

__device__ __forceinline__ float xor_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xor_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------ forward
template <int DKT, int DVT, int QB>
__global__ void __launch_bounds__(64) attn_fwd_k4_kernel(const PgAttnArgs a) {
  const int L = a.L, NB = (L + 16 * QB - 1) / (16 * QB);
  const K4Map mp = k4_map(a.N * a.heads, NB);
  if (mp.unit < 0) return;
  const int n = mp.unit / a.heads, h = mp.unit - n * a.heads;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * (4 * DKT) * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * (4 * DKT) * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * (16 * DVT) * L;
  float* op = a.o_out + (size_t)n * a.o_bs + (size_t)h * (16 * DVT) * L;
  float* lp = a.lse2_out + ((size_t)n * a.heads + h) * L;
  const int strict = a.strict;

  for (int pass = 0; pass < 2; ++pass) {
    const int b = pass == 0 ? mp.b0 : mp.b1;
    if (b < 0) break;
    const int q0 = b * (16 * QB);
    const int qlast = min(q0 + 16 * QB, L);             // groups at or beyond L are skipped
    const int nkt = (qlast - strict + 15) >> 4;         // key tiles this block can see
    float qf[QB][DKT];
#pragma unroll
    for (int qg = 0; qg < QB; ++qg)
#pragma unroll
      for (int c = 0; c < DKT; ++c)
        qf[qg][c] = qp[(size_t)(4 * c + g) * L + min(q0 + 16 * qg + j, L - 1)] * a.scale2;

    // ---- pass 1: row maxima (score MFMA only)
    float mx[QB];
#pragma unroll
    for (int qg = 0; qg < QB; ++qg) mx[qg] = NEG_BIG;
    {
      float kf[DKT];
#pragma unroll
      for (int c = 0; c < DKT; ++c) kf[c] = kp[(size_t)(4 * c + g) * L + j];
      for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt << 4;
        float kf_n[DKT];
        {
          const int kn = (kt + 1 < nkt) ? key0 + 16 : key0;
#pragma unroll
          for (int c = 0; c < DKT; ++c) kf_n[c] = kp[(size_t)(4 * c + g) * L + kn + j];
        }
#pragma unroll
        for (int qg = 0; qg < QB; ++qg) {
          const int qq0 = q0 + 16 * qg;
          if (key0 + strict > qq0 + 15 || qq0 >= L) continue;  // every key of the tile is masked for this group
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < DKT; ++c) s = MFMA4(kf[c], qf[qg][c], s);
          if (key0 + 15 + strict > qq0) {  // diagonal tile
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (key0 + 4 * g + r + strict > qq0 + j) s[r] = NEG_BIG;
          }
          mx[qg] = fmaxf(mx[qg], fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        }
#pragma unroll
        for (int c = 0; c < DKT; ++c) kf[c] = kf_n[c];
      }
    }
    float negm[QB];
#pragma unroll
    for (int qg = 0; qg < QB; ++qg) {
      const float m = xor_max(mx[qg]);
      negm[qg] = m > 0.5f * NEG_BIG ? -m : 0.f;  // empty row: any finite value (every p is masked)
    }

    // ---- pass 2: p = exp2(s - m), l += p, O^T += V^T P^T
    f32x4 O[QB][DVT];
    float l[QB];
#pragma unroll
    for (int qg = 0; qg < QB; ++qg) l[qg] = 0.f;
#pragma unroll
    for (int qg = 0; qg < QB; ++qg)
#pragma unroll
      for (int t = 0; t < DVT; ++t) O[qg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      float kf[DKT];
#pragma unroll
      for (int c = 0; c < DKT; ++c) kf[c] = kp[(size_t)(4 * c + g) * L + j];
      float4 vf[DVT];
#pragma unroll
      for (int t = 0; t < DVT; ++t)
        vf[t] = *reinterpret_cast<const float4*>(vp + (size_t)(16 * t + j) * L + 4 * g);
      for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt << 4;
        float kf_n[DKT];
        float4 vf_n[DVT];
        {
          const int kn = (kt + 1 < nkt) ? key0 + 16 : key0;  // clamped prefetch of the next tile
#pragma unroll
          for (int c = 0; c < DKT; ++c) kf_n[c] = kp[(size_t)(4 * c + g) * L + kn + j];
#pragma unroll
          for (int t = 0; t < DVT; ++t)
            vf_n[t] = *reinterpret_cast<const float4*>(vp + (size_t)(16 * t + j) * L + kn + 4 * g);
        }
#pragma unroll
        for (int qg = 0; qg < QB; ++qg) {
          const int qq0 = q0 + 16 * qg;
          if (key0 + strict > qq0 + 15 || qq0 >= L) continue;
          f32x4 s = f32x4{negm[qg], negm[qg], negm[qg], negm[qg]};
#pragma unroll
          for (int c = 0; c < DKT; ++c) s = MFMA4(kf[c], qf[qg][c], s);
          float p[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s[r]);
          if (key0 + 15 + strict > qq0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (key0 + 4 * g + r + strict > qq0 + j) p[r] = 0.f;
          }
          l[qg] += (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
          for (int t = 0; t < DVT; ++t) {
            O[qg][t] = MFMA4(vf[t].x, p[0], O[qg][t]);
            O[qg][t] = MFMA4(vf[t].y, p[1], O[qg][t]);
            O[qg][t] = MFMA4(vf[t].z, p[2], O[qg][t]);
            O[qg][t] = MFMA4(vf[t].w, p[3], O[qg][t]);
          }
        }
#pragma unroll
        for (int c = 0; c < DKT; ++c) kf[c] = kf_n[c];
#pragma unroll
        for (int t = 0; t < DVT; ++t) vf[t] = vf_n[t];
      }
    }
    // ---- normalise and write O^T[dv = 16t + 4g + r][query j], lse2
#pragma unroll
    for (int qg = 0; qg < QB; ++qg) {
      if (q0 + 16 * qg >= L) continue;
      const float lt = xor_sum(l[qg]);
      const float inv = lt > 0.f ? 1.f / lt : 0.f;
      const int qi = q0 + 16 * qg + j;
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(size_t)(16 * t + 4 * g + r) * L + qi] = O[qg][t][r] * inv;
      if (g == 0) lp[qi] = lt > 0.f ? -negm[qg] + log2f(lt) : LSE_EMPTY;
    }
  }
}

// ------------------------------------------------------------------------------------ dQ (+ delta)
template <int DKT, int DVT, int QB>
__global__ void __launch_bounds__(64) attn_dq_k4_kernel(const PgAttnArgs a) {
  const int L = a.L, NB = (L + 16 * QB - 1) / (16 * QB);
  const K4Map mp = k4_map(a.N * a.heads, NB);
  if (mp.unit < 0) return;
  const int n = mp.unit / a.heads, h = mp.unit - n * a.heads;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  constexpr int DK = 4 * DKT, DV = 16 * DVT, NS = 4 * DVT;  // dv K steps of 4
  constexpr int DKM = DKT >= 4 ? DKT / 4 : 1;               // 16-channel MFMA tiles of dQ (d_k >= 16)
  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * DK * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * DK * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * DV * L;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * DV * L;
  const float* dop = a.d_o + (size_t)n * a.do_bs + (size_t)h * DV * L;
  const float* lp = a.lse2_in + ((size_t)n * a.heads + h) * L;
  float* dlt = a.delta + ((size_t)n * a.heads + h) * L;
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * DK * L;
  const int strict = a.strict;

  for (int pass = 0; pass < 2; ++pass) {
    const int b = pass == 0 ? mp.b0 : mp.b1;
    if (b < 0) break;
    const int q0 = b * (16 * QB);
    const int qlast = min(q0 + 16 * QB, L);
    const int nkt = (qlast - strict + 15) >> 4;
    float qf[QB][DKT], nl[QB], dl[QB];
    float dof[QB][NS];  // B[k = dv][j = query] of dP^T = V dO^T: dO^T[dv = 4s + g][query j]
    float dq[QB][4];    // d_k = 4: VALU accumulation
    f32x4 dqm[QB][DKM];  // d_k >= 16: dQ^T[channel 16t + 4g + r][query j]
#pragma unroll
    for (int qg = 0; qg < QB; ++qg) {
      const int qi = min(q0 + 16 * qg + j, L - 1);
#pragma unroll
      for (int c = 0; c < DKT; ++c) qf[qg][c] = qp[(size_t)(4 * c + g) * L + qi] * a.scale2;
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        dof[qg][s] = dop[(size_t)(4 * s + g) * L + qi];
        acc = fmaf(dof[qg][s], op[(size_t)(4 * s + g) * L + qi], acc);
      }
      dl[qg] = xor_sum(acc);  // delta_j = sum_dv dO O
      if (g == 0 && q0 + 16 * qg < L) dlt[qi] = dl[qg];
      nl[qg] = -lp[qi];
#pragma unroll
      for (int d = 0; d < 4; ++d) dq[qg][d] = 0.f;
#pragma unroll
      for (int t = 0; t < DKM; ++t) dqm[qg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float kf[DKT];
    float vfa[NS];    // A[i = key][k = dv] of dP^T: V^T[dv = 4s + g][key j]
    float4 kr[DKT >= 4 ? DKM : 4];  // d_k = 4: K[d][keys 4g..4g+3]; d_k >= 16: K^T row 16t + j, keys 4g..4g+3
#pragma unroll
    for (int c = 0; c < DKT; ++c) kf[c] = kp[(size_t)(4 * c + g) * L + j];
#pragma unroll
    for (int s = 0; s < NS; ++s) vfa[s] = vp[(size_t)(4 * s + g) * L + j];
    if constexpr (DKT >= 4) {
#pragma unroll
      for (int t = 0; t < DKM; ++t) kr[t] = *reinterpret_cast<const float4*>(kp + (size_t)(16 * t + j) * L + 4 * g);
    } else {
#pragma unroll
      for (int d = 0; d < 4; ++d) kr[d] = *reinterpret_cast<const float4*>(kp + d * L + 4 * g);
    }
    for (int kt = 0; kt < nkt; ++kt) {
      const int key0 = kt << 4;
      const int kn = (kt + 1 < nkt) ? key0 + 16 : key0;
      float kf_n[DKT];
      float vfa_n[NS];
      float4 kr_n[DKT >= 4 ? DKM : 4];
#pragma unroll
      for (int c = 0; c < DKT; ++c) kf_n[c] = kp[(size_t)(4 * c + g) * L + kn + j];
#pragma unroll
      for (int s = 0; s < NS; ++s) vfa_n[s] = vp[(size_t)(4 * s + g) * L + kn + j];
      if constexpr (DKT >= 4) {
#pragma unroll
        for (int t = 0; t < DKM; ++t)
          kr_n[t] = *reinterpret_cast<const float4*>(kp + (size_t)(16 * t + j) * L + kn + 4 * g);
      } else {
#pragma unroll
        for (int d = 0; d < 4; ++d) kr_n[d] = *reinterpret_cast<const float4*>(kp + d * L + kn + 4 * g);
      }
#pragma unroll
      for (int qg = 0; qg < QB; ++qg) {
        const int qq0 = q0 + 16 * qg;
        if (key0 + strict > qq0 + 15 || qq0 >= L) continue;  // every key of the tile is masked for this group
        f32x4 s4 = f32x4{nl[qg], nl[qg], nl[qg], nl[qg]};
#pragma unroll
        for (int c = 0; c < DKT; ++c) s4 = MFMA4(kf[c], qf[qg][c], s4);
        f32x4 dp = f32x4{-dl[qg], -dl[qg], -dl[qg], -dl[qg]};
#pragma unroll
        for (int s = 0; s < NS; ++s) dp = MFMA4(vfa[s], dof[qg][s], dp);
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(s4[r]) * dp[r];
        if (key0 + 15 + strict > qq0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 4 * g + r + strict > qq0 + j) ds[r] = 0.f;
        }
        if constexpr (DKT >= 4) {  // dQ^T[16 channels][query] += K^T[16 channels x 4 keys] dS^T[4 keys x 16]
#pragma unroll
          for (int t = 0; t < DKM; ++t) {
            dqm[qg][t] = MFMA4(kr[t].x, ds[0], dqm[qg][t]);
            dqm[qg][t] = MFMA4(kr[t].y, ds[1], dqm[qg][t]);
            dqm[qg][t] = MFMA4(kr[t].z, ds[2], dqm[qg][t]);
            dqm[qg][t] = MFMA4(kr[t].w, ds[3], dqm[qg][t]);
          }
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d)
            dq[qg][d] += (ds[0] * kr[d].x + ds[1] * kr[d].y) + (ds[2] * kr[d].z + ds[3] * kr[d].w);
        }
      }
#pragma unroll
      for (int c = 0; c < DKT; ++c) kf[c] = kf_n[c];
#pragma unroll
      for (int s = 0; s < NS; ++s) vfa[s] = vfa_n[s];
#pragma unroll
      for (int d = 0; d < (DKT >= 4 ? DKM : 4); ++d) kr[d] = kr_n[d];
    }
#pragma unroll
    for (int qg = 0; qg < QB; ++qg) {
      if (q0 + 16 * qg >= L) continue;
      if constexpr (DKT >= 4) {
#pragma unroll
        for (int t = 0; t < DKM; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dqp[(size_t)(16 * t + 4 * g + r) * L + q0 + 16 * qg + j] = dqm[qg][t][r] * a.scale;
      } else {
        // lanes (0..3, j) hold partial sums over their keys; lane (g, j) writes channel d = g
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float t = xor_sum(dq[qg][d]);
          if (d == g) mine = t;
        }
        dqp[g * L + q0 + 16 * qg + j] = mine * a.scale;
      }
    }
  }
}

// ------------------------------------------------------------------------------------ dK, dV
template <int DKT, int DVT, int QB>
__global__ void __launch_bounds__(64) attn_dkv_k4_kernel(const PgAttnArgs a) {
  const int L = a.L, NB = (L + 16 * QB - 1) / (16 * QB);
  const K4Map mp = k4_map(a.N * a.heads, NB);
  if (mp.unit < 0) return;
  const int n = mp.unit / a.heads, h = mp.unit - n * a.heads;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  constexpr int DK = 4 * DKT, DV = 16 * DVT, NS = 4 * DVT;
  constexpr int DKM = DKT >= 4 ? DKT / 4 : 1;
  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * DK * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * DK * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * DV * L;
  const float* dop = a.d_o + (size_t)n * a.do_bs + (size_t)h * DV * L;
  const float* lp = a.lse2_in + ((size_t)n * a.heads + h) * L;
  const float* dlt = a.delta + ((size_t)n * a.heads + h) * L;
  float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * DK * L;
  float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * DV * L;
  const int strict = a.strict;
  const int nqt = L >> 4;

  for (int pass = 0; pass < 2; ++pass) {
    // key-owner: block 0 streams the most queries, so the pair is (NB-1-b0, NB-1-b1) mirrored
    const int bb = pass == 0 ? mp.b0 : mp.b1;
    if (bb < 0) break;
    const int k0 = (NB - 1 - bb) * (16 * QB);
    float kfb[QB][DKT];  // B[k = d][j = key] of S = Q K^T, per 16-key group
    float vfb[QB][NS];   // B[k = dv][j = key] of dP = dO V^T
    f32x4 dV[QB][DVT];
    float dk[QB][4];     // d_k = 4: VALU accumulation
    f32x4 dkm[QB][DKM];  // d_k >= 16: dK^T[channel 16t + 4g + r][key j]
#pragma unroll
    for (int kg = 0; kg < QB; ++kg) {
      const int ki = min(k0 + 16 * kg + j, L - 1);
#pragma unroll
      for (int c = 0; c < DKT; ++c) kfb[kg][c] = kp[(size_t)(4 * c + g) * L + ki] * a.scale2;
#pragma unroll
      for (int s = 0; s < NS; ++s) vfb[kg][s] = vp[(size_t)(4 * s + g) * L + ki];
#pragma unroll
      for (int t = 0; t < DVT; ++t) dV[kg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 4; ++d) dk[kg][d] = 0.f;
#pragma unroll
      for (int t = 0; t < DKM; ++t) dkm[kg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int qt0 = (k0 + strict) >> 4;  // first query tile that sees a key of this block

#define K4_LOAD(QQ, QFA, DOFA, DOFT, QR, NL, ND)                                                   \
  {                                                                                                 \
    _Pragma("unroll") for (int c = 0; c < DKT; ++c) QFA[c] = qp[(size_t)(4 * c + g) * L + (QQ) + j]; \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) DOFA[s] = dop[(size_t)(4 * s + g) * L + (QQ) + j]; \
    _Pragma("unroll") for (int t = 0; t < DVT; ++t)                                                 \
        DOFT[t] = *reinterpret_cast<const float4*>(dop + (size_t)(16 * t + j) * L + (QQ) + 4 * g);   \
    if constexpr (DKT >= 4) {                                                                       \
      _Pragma("unroll") for (int t = 0; t < DKM; ++t)                                               \
          QR[t] = *reinterpret_cast<const float4*>(qp + (size_t)(16 * t + j) * L + (QQ) + 4 * g);    \
    } else {                                                                                        \
      _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                 \
          QR[d] = *reinterpret_cast<const float4*>(qp + d * L + (QQ) + 4 * g);                       \
    }                                                                                               \
    NL = *reinterpret_cast<const float4*>(lp + (QQ) + 4 * g);                                        \
    ND = *reinterpret_cast<const float4*>(dlt + (QQ) + 4 * g);                                       \
  }
    float qfa[DKT], dofa[NS];
    float4 doft[DVT], qr[DKT >= 4 ? DKM : 4], nl4, nd4;
    K4_LOAD(min(qt0, nqt - 1) << 4, qfa, dofa, doft, qr, nl4, nd4)
    for (int qt = qt0; qt < nqt; ++qt) {
      const int qq0 = qt << 4;
      const int qn = (qt + 1 < nqt) ? qq0 + 16 : qq0;
      float qfa_n[DKT], dofa_n[NS];
      float4 doft_n[DVT], qr_n[DKT >= 4 ? DKM : 4], nl4_n, nd4_n;
      K4_LOAD(qn, qfa_n, dofa_n, doft_n, qr_n, nl4_n, nd4_n)
#pragma unroll
      for (int kg = 0; kg < QB; ++kg) {
        const int kk0 = k0 + 16 * kg;
        if (kk0 + strict > qq0 + 15 || kk0 >= L) continue;  // no query of the tile sees a key of this group
        // S[query 4g + r][key j] - lse2[query]
        f32x4 s4 = f32x4{-nl4.x, -nl4.y, -nl4.z, -nl4.w};
#pragma unroll
        for (int c = 0; c < DKT; ++c) s4 = MFMA4(qfa[c], kfb[kg][c], s4);
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s4[r]);
        if (kk0 + 15 + strict > qq0) {  // diagonal tile
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kk0 + j + strict > qq0 + 4 * g + r) p[r] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < DVT; ++t) {  // dV^T[dv][key] += dO^T[dv][query] P[query][key]
          dV[kg][t] = MFMA4(doft[t].x, p[0], dV[kg][t]);
          dV[kg][t] = MFMA4(doft[t].y, p[1], dV[kg][t]);
          dV[kg][t] = MFMA4(doft[t].z, p[2], dV[kg][t]);
          dV[kg][t] = MFMA4(doft[t].w, p[3], dV[kg][t]);
        }
        f32x4 dp = f32x4{-nd4.x, -nd4.y, -nd4.z, -nd4.w};
#pragma unroll
        for (int s = 0; s < NS; ++s) dp = MFMA4(dofa[s], vfb[kg][s], dp);
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = p[r] * dp[r];
        if constexpr (DKT >= 4) {  // dK^T[16 channels][key] += Q^T[16 channels x 4 queries] dS[4 queries x 16 keys]
#pragma unroll
          for (int t = 0; t < DKM; ++t) {
            dkm[kg][t] = MFMA4(qr[t].x, ds[0], dkm[kg][t]);
            dkm[kg][t] = MFMA4(qr[t].y, ds[1], dkm[kg][t]);
            dkm[kg][t] = MFMA4(qr[t].z, ds[2], dkm[kg][t]);
            dkm[kg][t] = MFMA4(qr[t].w, ds[3], dkm[kg][t]);
          }
        } else {
#pragma unroll
          for (int d = 0; d < 4; ++d)
            dk[kg][d] += (ds[0] * qr[d].x + ds[1] * qr[d].y) + (ds[2] * qr[d].z + ds[3] * qr[d].w);
        }
      }
      nl4 = nl4_n; nd4 = nd4_n;
#pragma unroll
      for (int c = 0; c < DKT; ++c) qfa[c] = qfa_n[c];
#pragma unroll
      for (int s = 0; s < NS; ++s) dofa[s] = dofa_n[s];
#pragma unroll
      for (int t = 0; t < DVT; ++t) doft[t] = doft_n[t];
#pragma unroll
      for (int d = 0; d < (DKT >= 4 ? DKM : 4); ++d) qr[d] = qr_n[d];
    }
#undef K4_LOAD
#pragma unroll
    for (int kg = 0; kg < QB; ++kg) {
      if (k0 + 16 * kg >= L) continue;
      const int ki = k0 + 16 * kg + j;
      if constexpr (DKT >= 4) {
#pragma unroll
        for (int t = 0; t < DKM; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) dkp[(size_t)(16 * t + 4 * g + r) * L + ki] = dkm[kg][t][r] * a.scale;
      } else {
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float t = xor_sum(dk[kg][d]);
          if (d == g) mine = t;
        }
        dkp[g * L + ki] = mine * a.scale;
      }
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dvp[(size_t)(16 * t + 4 * g + r) * L + ki] = dV[kg][t][r];
    }
  }
}

// ------------------------------------------------------------------------------------ fused backward (d_k = 4)
// dQ, dK, dV in ONE pass (round 4): the two-kernel backward evaluates the score tile, exp2 and dP = dO V^T twice
// (dq kernel: 4 d_k + 2 d_v, dkv kernel: 4 d_k + 4 d_v FLOP per pair; fused: 6 d_k + 4 d_v, every product once —
// 152 against 224 FLOP per pair at d_v = 32). Key-owner orientation as attn_dkv_k4_kernel (a wave owns a block of
// keys, the queries stream; dV / dK accumulate in registers). dQ contracts over KEYS, which sit on the lane axis of
// the score tile S[query 4g + r][key j]: each 16 x 16 dS tile is transposed through a 1.25 KB per-wave LDS scratch
// (one ds_write_b128 + four ds_read_b32; rows of 20 floats: conflict free both ways), multiplied by K^T float4s on
// the VALU (d_k = 4: an MFMA tile would be 75 % padding), summed over the block's key groups in registers and over
// the four lane groups by two shuffles, and deposited with ONE fp32 atomic per lane and query tile into dq. delta =
// sum_dv dO * O comes from a small pre-pass (attn_delta_k4_kernel), which also zeroes dq. dQ's summation order
// over key blocks depends on the schedule: pg_attn_fused_bwd(0) / ops.set_deterministic(True) keep the two-kernel
// backward for bit-reproducible gradients, as for d_k = d_v = 4.
template <int DVT>
__global__ void __launch_bounds__(256) attn_delta_k4_kernel(const PgAttnArgs a) {
  const int unit = blockIdx.y, n = unit / a.heads, h = unit - n * a.heads;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= a.L) return;
  constexpr int DV = 16 * DVT;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * DV * a.L + q;
  const float* dop = a.d_o + (size_t)n * a.do_bs + (size_t)h * DV * a.L + q;
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < DV; ++c) acc = fmaf(dop[(size_t)c * a.L], op[(size_t)c * a.L], acc);
  a.delta[((size_t)n * a.heads + h) * a.L + q] = acc;
  // dQ is deposited with atomics by the fused kernel, so this pre-pass also zeroes it. (Not a memset node: with
  // hipMemset2DAsync here, a torch-captured graph of forward + backward returned the right dQ on its first launch and
  // dQ off by a constant on every later one — ROCm 7.2, tools/exp/attn_graph_check.py; a lone memset node replays
  // correctly, tools/exp/memset_graph_test.hip, so the node's interplay with its neighbours is what fails.)
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * 4 * a.L + q;
#pragma unroll
  for (int c = 0; c < 4; ++c) dqp[(size_t)c * a.L] = 0.f;
}

template <int DVT, int QB>
__global__ void __launch_bounds__(64) attn_bwd_k4_kernel(const PgAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float scr[16 * 20];
  const int L = a.L, NB = (L + 16 * QB - 1) / (16 * QB);
  const K4Map mp = k4_map(a.N * a.heads, NB);
  if (mp.unit < 0) return;
  const int n = mp.unit / a.heads, h = mp.unit - n * a.heads;
  const int lane = threadIdx.x, g = lane >> 4, j = lane & 15;
  constexpr int DV = 16 * DVT, NS = 4 * DVT;
  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * DV * L;
  const float* dop = a.d_o + (size_t)n * a.do_bs + (size_t)h * DV * L;
  const float* lp = a.lse2_in + ((size_t)n * a.heads + h) * L;
  const float* dlt = a.delta + ((size_t)n * a.heads + h) * L;
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * 4 * L;
  float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * 4 * L;
  float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * DV * L;
  const int strict = a.strict;
  const int nqt = L >> 4;
  float* scr_w = scr + j * 20 + 4 * g;   // this lane's dS values: row = its key, columns = its four queries
  const float* scr_r = scr + 4 * g * 20 + j;  // transposed: rows = keys 4g .. 4g+3, column = query j

  for (int pass = 0; pass < 2; ++pass) {
    const int bb = pass == 0 ? mp.b0 : mp.b1;
    if (bb < 0) break;
    const int k0 = (NB - 1 - bb) * (16 * QB);
    float kfb[QB], vfb[QB][NS];
    float4 kr[QB][4];     // K^T row d at keys 4g .. 4g+3 of the group (unscaled: dQ is scaled once at the end)
    f32x4 dV[QB][DVT];
    float dk[QB][4];
#pragma unroll
    for (int kg = 0; kg < QB; ++kg) {
      const int ki = min(k0 + 16 * kg + j, L - 1);
      kfb[kg] = kp[(size_t)g * L + ki] * a.scale2;
#pragma unroll
      for (int s = 0; s < NS; ++s) vfb[kg][s] = vp[(size_t)(4 * s + g) * L + ki];
      const int kq = min(k0 + 16 * kg + 4 * g, L - 4);  // (L % 16 == 0: a group beyond L is skipped below)
#pragma unroll
      for (int d = 0; d < 4; ++d) kr[kg][d] = *reinterpret_cast<const float4*>(kp + (size_t)d * L + kq);
#pragma unroll
      for (int t = 0; t < DVT; ++t) dV[kg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 4; ++d) dk[kg][d] = 0.f;
    }
    const int qt0 = (k0 + strict) >> 4;  // first query tile that sees a key of this block

#define K4B_LOAD(QQ, QFA, DOFA, DOFT, QR, NL, ND)                                                   \
  {                                                                                                 \
    QFA = qp[(size_t)g * L + (QQ) + j];                                                             \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) DOFA[s] = dop[(size_t)(4 * s + g) * L + (QQ) + j]; \
    _Pragma("unroll") for (int t = 0; t < DVT; ++t)                                                 \
        DOFT[t] = *reinterpret_cast<const float4*>(dop + (size_t)(16 * t + j) * L + (QQ) + 4 * g);   \
    _Pragma("unroll") for (int d = 0; d < 4; ++d)                                                   \
        QR[d] = *reinterpret_cast<const float4*>(qp + (size_t)d * L + (QQ) + 4 * g);                 \
    NL = *reinterpret_cast<const float4*>(lp + (QQ) + 4 * g);                                        \
    ND = *reinterpret_cast<const float4*>(dlt + (QQ) + 4 * g);                                       \
  }
    float qfa, dofa[NS];
    float4 doft[DVT], qr[4], nl4, nd4;
    K4B_LOAD(min(qt0, nqt - 1) << 4, qfa, dofa, doft, qr, nl4, nd4)
    for (int qt = qt0; qt < nqt; ++qt) {
      const int qq0 = qt << 4;
      const int qn = (qt + 1 < nqt) ? qq0 + 16 : qq0;
      float qfa_n, dofa_n[NS];
      float4 doft_n[DVT], qr_n[4], nl4_n, nd4_n;
      K4B_LOAD(qn, qfa_n, dofa_n, doft_n, qr_n, nl4_n, nd4_n)
      float dqa[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kg = 0; kg < QB; ++kg) {
        const int kk0 = k0 + 16 * kg;
        if (kk0 + strict > qq0 + 15 || kk0 >= L) continue;  // no query of the tile sees a key of this group
        f32x4 s4 = f32x4{-nl4.x, -nl4.y, -nl4.z, -nl4.w};
        s4 = MFMA4(qfa, kfb[kg], s4);  // S[query 4g + r][key j] - lse2[query]
        float p[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(s4[r]);
        if (kk0 + 15 + strict > qq0) {  // diagonal tile
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kk0 + j + strict > qq0 + 4 * g + r) p[r] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < DVT; ++t) {  // dV^T[dv][key] += dO^T[dv][query] P[query][key]
          dV[kg][t] = MFMA4(doft[t].x, p[0], dV[kg][t]);
          dV[kg][t] = MFMA4(doft[t].y, p[1], dV[kg][t]);
          dV[kg][t] = MFMA4(doft[t].z, p[2], dV[kg][t]);
          dV[kg][t] = MFMA4(doft[t].w, p[3], dV[kg][t]);
        }
        f32x4 dp = f32x4{-nd4.x, -nd4.y, -nd4.z, -nd4.w};
#pragma unroll
        for (int s = 0; s < NS; ++s) dp = MFMA4(dofa[s], vfb[kg][s], dp);
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[r] = p[r] * dp[r];
#pragma unroll
        for (int d = 0; d < 4; ++d)  // dK[key j][d] += dS^T Q: contraction over the tile's queries = this lane's registers
          dk[kg][d] += (ds[0] * qr[d].x + ds[1] * qr[d].y) + (ds[2] * qr[d].z + ds[3] * qr[d].w);
        // dQ[query][d] += dS K: transpose the tile (a wave's LDS operations execute in order: no barrier needed)
        *reinterpret_cast<f32x4*>(scr_w) = ds;
        asm volatile("" ::: "memory");
        float tr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) tr[r] = scr_r[r * 20];  // dS[query j][key 4g + r]
        asm volatile("" ::: "memory");
#pragma unroll
        for (int d = 0; d < 4; ++d)
          dqa[d] += (tr[0] * kr[kg][d].x + tr[1] * kr[kg][d].y) + (tr[2] * kr[kg][d].z + tr[3] * kr[kg][d].w);
      }
      {  // sum the four key quads of every query (lane groups), lane (g, j) deposits channel g of query qq0 + j
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float t = xor_sum(dqa[d]);
          if (d == g) mine = t;
        }
        if (qq0 + j < L) atomicAdd(dqp + (size_t)g * L + qq0 + j, mine * a.scale);
      }
      qfa = qfa_n; nl4 = nl4_n; nd4 = nd4_n;
#pragma unroll
      for (int s = 0; s < NS; ++s) dofa[s] = dofa_n[s];
#pragma unroll
      for (int t = 0; t < DVT; ++t) doft[t] = doft_n[t];
#pragma unroll
      for (int d = 0; d < 4; ++d) qr[d] = qr_n[d];
    }
#undef K4B_LOAD
#pragma unroll
    for (int kg = 0; kg < QB; ++kg) {
      if (k0 + 16 * kg >= L) continue;
      const int ki = k0 + 16 * kg + j;
      {
        float mine = 0.f;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float t = xor_sum(dk[kg][d]);
          if (d == g) mine = t;
        }
        dkp[(size_t)g * L + ki] = mine * a.scale;
      }
#pragma unroll
      for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dvp[(size_t)(16 * t + 4 * g + r) * L + ki] = dV[kg][t][r];
    }
  }
}

template <int DKT, int DVT, int QB>
void k4_launch(int which, const PgAttnArgs& a, dim3 grid, hipStream_t st) {
  if (which == PG_ATTN_FWD)
    hipLaunchKernelGGL((attn_fwd_k4_kernel<DKT, DVT, QB>), grid, dim3(64), 0, st, a);
  else if (which == PG_ATTN_DQ)
    hipLaunchKernelGGL((attn_dq_k4_kernel<DKT, DVT, QB>), grid, dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL((attn_dkv_k4_kernel<DKT, DVT, QB>), grid, dim3(64), 0, st, a);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// Returns 1 if these kernels took the launch: d_k = 4 with d_v in {16, 32, 64}, or d_k in {16, 32, 64} with d_v in
// {16, 32, 64}; L % 16 == 0; 16-byte aligned planes; else 0. (Round 4: everything beyond (4, 16 / 32) and (32, 32)
// — the shapes that used to run on VALU kernels spilling up to 1 755 registers.)
int pg_attn_k4_launch(int which, const PgAttnArgs& a, hipStream_t st) {
  const int dk = a.dk_dim, dv = a.dv_dim;
  const bool dv_ok = dv == 16 || dv == 32 || dv == 64;
  const bool small_k = dk == 4 && dv_ok;
  const bool big_k = (dk == 16 || dk == 32 || dk == 64) && dv_ok;
  if (!(small_k || big_k) || (a.L % 16) != 0 || a.L < 16) return 0;
  if (which == PG_ATTN_BWD && !(small_k && dv <= 32)) return 0;  // the fused backward exists for d_k = 4, d_v = 16 / 32
  if (which > PG_ATTN_BWD) return 0;
  if ((a.q_bs | a.k_bs | a.v_bs | a.o_bs) % 4 != 0) return 0;
  if (!al16(a.k) || !al16(a.v) || !al16(a.q)) return 0;
  if (which != PG_ATTN_FWD) {
    if ((a.do_bs | a.dq_bs | a.dk_bs | a.dv_bs) % 4 != 0) return 0;
    if (!al16(a.d_o) || !al16(a.lse2_in) || !al16(a.delta)) return 0;
  }
  const int units = a.N * a.heads;
  if (which == PG_ATTN_BWD) {
    // delta pre-pass (which also zeroes dq: the fused kernel deposits with atomics), then the fused kernel on 32-key blocks
    if ((a.dq_bs % 4) != 0 || (reinterpret_cast<uintptr_t>(a.dq) & 15) != 0) return 0;
    const dim3 dgrid((unsigned)((a.L + 255) / 256), (unsigned)units);
    if (dv == 32) hipLaunchKernelGGL((attn_delta_k4_kernel<2>), dgrid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_delta_k4_kernel<1>), dgrid, dim3(256), 0, st, a);
    constexpr int FQB = 2;
    const int NBf = (a.L + 16 * FQB - 1) / (16 * FQB);
    const int npf = (NBf + 1) / 2;
    const dim3 fgrid((unsigned)(((units + 63) / 64) * 64 * npf));
    if (dv == 32) hipLaunchKernelGGL((attn_bwd_k4_kernel<2, FQB>), fgrid, dim3(64), 0, st, a);
    else hipLaunchKernelGGL((attn_bwd_k4_kernel<1, FQB>), fgrid, dim3(64), 0, st, a);
    return 1;
  }
  // A lone wave keeps its SIMD's matrix pipe ~50 % busy (score -> exp -> P.V is one dependent chain
  // per 16-query group), so below ~3 waves per SIMD the blocks are halved to 32 rows: twice the
  // waves, each half as long (measured at batch 128: 1 wave per SIMD ran 2x over the MFMA bound).
  // Rows per block otherwise follow the register file: the resident operands (q / k / v / dO fragments of every
  // 16-row group of the block) must fit — 32 rows from d_k = 16 on, 16 rows when d_k or d_v is 64.
  static const int force_qb = []() { const char* e = PG_AB_ENV("PG_ATTN_K4_QB"); return e ? atoi(e) : 0; }();
  int qb;
  if (small_k && dv <= 32) {
    qb = ((long)units * (((a.L + 63) / 64 + 1) / 2) >= 3 * 1024) ? 4 : 2;
    if (force_qb == 2 || force_qb == 4) qb = force_qb;
  } else if (small_k) {
    qb = 2;
  } else {
    qb = (dk == 64 || dv == 64) ? 1 : 2;
  }
  const int NB = (a.L + 16 * qb - 1) / (16 * qb);
  const int npair = (NB + 1) / 2;
  // units in groups of 64 (8 per XCD); every group has 8 * 8 * npair workgroups
  const int groups = (units + 63) / 64;
  dim3 grid((unsigned)(groups * 64 * npair));
  if (small_k) {
    if (dv == 64) k4_launch<1, 4, 2>(which, a, grid, st);
    else if (dv == 32) { if (qb == 4) k4_launch<1, 2, 4>(which, a, grid, st); else k4_launch<1, 2, 2>(which, a, grid, st); }
    else { if (qb == 4) k4_launch<1, 1, 4>(which, a, grid, st); else k4_launch<1, 1, 2>(which, a, grid, st); }
  } else if (dk == 16) {
    if (dv == 16) k4_launch<4, 1, 2>(which, a, grid, st);
    else if (dv == 32) k4_launch<4, 2, 2>(which, a, grid, st);
    else k4_launch<4, 4, 1>(which, a, grid, st);
  } else if (dk == 32) {
    if (dv == 16) k4_launch<8, 1, 2>(which, a, grid, st);
    else if (dv == 32) k4_launch<8, 2, 2>(which, a, grid, st);
    else k4_launch<8, 4, 1>(which, a, grid, st);
  } else {
    if (dv == 16) k4_launch<16, 1, 1>(which, a, grid, st);
    else if (dv == 32) k4_launch<16, 2, 1>(which, a, grid, st);
    else k4_launch<16, 4, 1>(which, a, grid, st);
  }
  return 1;
}
