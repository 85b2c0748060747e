// attention_mfma.hip — matrix-core causal attention (forward, dQ, dK/dV) for d_k = d_v = 4, gfx950.
//
// ImageGPT (16 embed / 4 heads, BASELINE.json configs[1]) has d_k = d_v = 4: the contraction depth of
// v_mfma_f32_16x16x4_f32 and the tile edge of v_mfma_f32_4x4x1_16b_f32. Measured on MI355X
// (tools/exp/ubench.hip): an fp32 MFMA and VALU work do NOT overlap on a SIMD
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0; MFMA + VALU streams cost the sum of their parts), so the matrix
// pipe buys cheaper multiply-adds, not a second pipe: a 16x16x4 fp32 tile costs 33 cycles for 4 scores
// per lane (16 v_fma = 58), a 4x4x1 costs 10.5 for 4 output FMAs per lane (14.4), v_exp_f32 10-12.
// A bf16 16x16x32 MFMA costs 17.7 cycles AND runs under other waves' v_exp: the score products of
// the forward and dQ kernels therefore go through it as exact three-way bf16 splits ("bf16x3",
// below) — fp32-level accuracy, no precision is given up anywhere on this path.
//
// One 16 keys x 16 queries tile, forward:
//   S^T - m = K Q^T - m    one 16x16 MFMA: A = keys streamed from LDS, B = queries held in VGPRs,
//                          the running max m (per query) folded into the contraction
//     D layout: lane (j = lane&15, g = lane>>4), VGPR r  <->  key 4g+r, query j
//   P = exp2(S^T - m)      4 x v_exp_f32 per lane
//   O += P^T V             4 x v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products):
//     block b = lane>>2 = (g, query quad qb); step r: A_b[i'] = P[key 4g+r][query 4qb+i'] — VGPR r
//     of the S^T tile AS IT IS, no cross-lane movement; B_b[j'] = V[key 4g+r][j'] — component r of
//     one ds_read_b128 from the V^T plane; D_b[i'][j'] = O partial [query 4qb+i'][channel j'].
//   No lane is padding in either instruction. The four key subsets g of a query keep lane-local
//   sums / outputs (same m) and are added once per 64-query block.
// m is updated LAZILY: the first tile sets every query's m to its exact maximum; afterwards m moves
// only when a tile's probability sum shows that some score ran more than ~2^8 above it (one
// wave-uniform test per 4 groups).
// dQ and dK/dV have the same shape: S and dP^T = V dO^T with -lse2 resp. -delta folded in, so
// exp2(D) = P and D = dP - delta come straight out of the matrix pipe, then dQ += dS K, or
// dV += P^T dO and dK += dS^T Q as 4x4x1 outer-product accumulations. (dK/dV uses the bf16x3 tile for S
// only: chunks of both q and dO on its streamed side would need 96 B per query and no longer fit two
// workgroups per CU, and resident bf16x3 operands for k and v would not fit 128 registers; dP stays on
// the fp32 16x16x4 tile, -lse2 / -delta in the C operands.)
// A wave works on a 64-row block = four 16-row groups at a time (one K/V fragment feeds four
// independent MFMA/exp chains). Each wave walks a host-made longest-processing-time list of blocks;
// workgroups have 4 (forward) or 8 (dQ, dK/dV) waves so that a CU's four SIMDs carry equal loads.
// Measured (N=1024, 4 heads, L=784): fwd 0.37 ms, dQ 0.42 ms, dK/dV 0.52 ms vs 0.62 / 0.65 / 0.69 ms
// for the VALU row-owner kernels; ~0.1 ms of each is staging, per-block set-up and the diagonal.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "attention_args.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float NEG_BIG = -1.0e30f;
constexpr float POS_BIG = 1.0e30f;
constexpr float PSUM_TH = 1024.0f;  // a 4-key probability sum above this (some p > 2^8) triggers a rescale

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#define MFMA16B(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16((A), (B), (C), 0, 0, 0)
#define MFMA4(A, B, C) __builtin_amdgcn_mfma_f32_4x4x1f32((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// out[i] = value of lane 4*(lane/4) + i, the i-th lane of this lane's 4x4x1 block (DPP quad_perm)
__device__ __forceinline__ void quad_all(float x, float (&out)[4]) {
  const int v = __float_as_int(x);
  out[0] = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x00, 0xf, 0xf, true));
  out[1] = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x55, 0xf, 0xf, true));
  out[2] = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0xaa, 0xf, 0xf, true));
  out[3] = __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0xff, 0xf, 0xf, true));
}

// sum over the four key/query subsets g = lane>>4
__device__ __forceinline__ float gsum(float x) {
  x += __shfl_xor(x, 16, 64);
  x += __shfl_xor(x, 32, 64);
  return x;
}

// ---- fp32 products on the bf16 matrix pipe ("bf16x3") --------------------------------------------
// x = h + m + l with h, m, l bf16 (8 + 8 + 8 significand bits: |x - h - m - l| <= 2^-24 |x|).
// A 4-deep fp32 dot product x.y + c then is one v_mfma_f32_16x16x32_bf16 (K = 32 = 8 slots of 4):
//   lane group kg = lane>>4 | streamed operand (LDS chunk) | resident operand (VGPRs)
//        0                  | [ yh | ym ]                  | [ xh | xh ]      yh.xh + ym.xh
//        1                  | [ yh | ym ]  (same chunk)    | [ xm | xm ]      yh.xm + ym.xm
//        2                  | [ yl | yh ]                  | [ xh | xl ]      yl.xh + yh.xl
//        3                  | [ 1 1 1 0 | 0 ]  (constant)  | [ ch cm cl 0 | 0 ]   c = ch + cm + cl
// Every bf16 x bf16 product is exact in the fp32 accumulator; the dropped terms (ym.xl, yl.xm,
// yl.xl) are <= 3 * 2^-24 |x||y| — fp32 rounding level. The per-column constant c (minus the running
// max, minus lse2, minus delta) rides in the contraction, so no C operand registers are needed.
// Measured (tools/exp/ubench.hip): 17.7 cycles per instruction vs 33 for v_mfma_f32_16x16x4_f32,
// and unlike the fp32 MFMA it overlaps with v_exp_f32 issued by other waves of the SIMD.
struct Split3 { __bf16 h, m, l; };
__device__ __forceinline__ Split3 split3(float x) {
  Split3 s;
  s.h = (__bf16)x;
  const float r1 = x - (float)s.h;
  s.m = (__bf16)r1;
  s.l = (__bf16)(r1 - (float)s.m);
  return s;
}

// resident operand of one row x[0..3] (+ the row's additive constant c) for this lane's group kg
__device__ __forceinline__ bf16x8 resident_operand(const float (&x)[4], float c, int kg) {
  bf16x8 v;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const Split3 s = split3(x[dd]);
    v[dd] = kg == 1 ? s.m : s.h;
    v[4 + dd] = kg == 0 ? s.h : (kg == 1 ? s.m : s.l);
  }
  const Split3 cc = split3(c);
  if (kg == 3) {
    v[0] = cc.h; v[1] = cc.m; v[2] = cc.l; v[3] = (__bf16)0.f;
    v[4] = v[5] = v[6] = v[7] = (__bf16)0.f;
  }
  return v;
}

// replace the additive constant of a resident operand (lanes of group 3 only; branch-free: the
// operand is rewritten in place as two selected dwords)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned int pack_bf16(__bf16 lo, __bf16 hi) {
  return (unsigned int)__builtin_bit_cast(unsigned short, lo) |
         ((unsigned int)__builtin_bit_cast(unsigned short, hi) << 16);
}
__device__ __forceinline__ void set_constant(bf16x8& v, float c, int kg) {
  const Split3 cc = split3(c);
  u32x4 w = __builtin_bit_cast(u32x4, v);
  w[0] = kg == 3 ? pack_bf16(cc.h, cc.m) : w[0];
  w[1] = kg == 3 ? pack_bf16(cc.l, (__bf16)0.f) : w[1];
  v = __builtin_bit_cast(bf16x8, w);
}

// streamed chunks of one row y[0..3]: c0[row] = [yh | ym], c2[row] = [yl | yh] (8 bf16 = 16 bytes each)
__device__ __forceinline__ void put_chunks(bf16x8* __restrict__ c0, bf16x8* __restrict__ c2, int row,
                                           const float (&y)[4]) {
  bf16x8 a, b;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const Split3 s = split3(y[dd]);
    a[dd] = s.h; a[4 + dd] = s.m;
    b[dd] = s.l; b[4 + dd] = s.h;
  }
  c0[row] = a;
  c2[row] = b;
}

// 4 channel values of row m (0 beyond L); the load itself is unconditional (clamped address), so the
// loads of a whole staging pass are in flight together
__device__ __forceinline__ void load_row4(float (&x)[4], const float* __restrict__ src, int L, int m) {
  const int mc = m < L ? m : L - 1;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) x[dd] = src[(size_t)dd * L + mc];
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) x[dd] = m < L ? x[dd] : 0.f;
}

// the constant chunk [1 1 1 0 | 0 0 0 0] read by lane group 3
__device__ __forceinline__ bf16x8 ones_chunk() {
  bf16x8 v;
  v[0] = v[1] = v[2] = (__bf16)1.f;
  v[3] = v[4] = v[5] = v[6] = v[7] = (__bf16)0.f;
  return v;
}

// max over the four key subsets g = lane>>4
__device__ __forceinline__ float gmax(float x) {
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}

// Store a lane's 4 consecutive rows (row0..row0+3) of channel plane `plane`.
__device__ __forceinline__ void store_rows4(float* __restrict__ plane, int L, int row0, bool vec,
                                            float v0, float v1, float v2, float v3) {
  if (vec) {
    if (row0 < L) *reinterpret_cast<float4*>(plane + row0) = make_float4(v0, v1, v2, v3);
  } else {
    if (row0 < L) plane[row0] = v0;
    if (row0 + 1 < L) plane[row0 + 1] = v1;
    if (row0 + 2 < L) plane[row0 + 2] = v2;
    if (row0 + 3 < L) plane[row0 + 3] = v3;
  }
}

template <int V> using I = std::integral_constant<int, V>;
template <bool V> using B = std::integral_constant<bool, V>;

struct Ids {
  int wave, lane, qi, g, jc, qb4;
};
__device__ __forceinline__ Ids ids() {
  Ids d;
  d.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  d.lane = threadIdx.x & 63;
  d.qi = d.lane & 15;               // MFMA16: A row / B column / D column owned by this lane
  d.g = d.lane >> 4;                // MFMA16: contraction index (A, B) / D row group
  d.jc = d.lane & 3;                // MFMA4: lane within the 4x4 block (A row, B and D column)
  d.qb4 = ((d.lane >> 2) & 3) * 4;  // MFMA4: first of the block's four D^T columns (rows of O)
  return d;
}

// ------------------------------------------------------------------------------ forward
// LDS per workgroup: K chunks c0/c2 [Lp] x 16 B, V^T planes [4][Lp] fp32, the ones chunk.
__global__ void __launch_bounds__(512) attn_fwd_m44_kernel(const PgAttnArgs a) {
  extern __shared__ float4 lds4[];
  const int Lp = a.lp;
  bf16x8* kc0 = reinterpret_cast<bf16x8*>(lds4);
  bf16x8* kc2 = kc0 + Lp;
  float* vt = reinterpret_cast<float*>(kc2 + Lp);  // V^T [4][Lp]
  bf16x8* ones = reinterpret_cast<bf16x8*>(vt + 4 * Lp);
  const Ids d = ids();
  const int qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  float* op = a.o_out + (size_t)n * a.o_bs + (size_t)h * 4 * L;
  float* lsep = a.lse2_out + ((size_t)n * a.heads + h) * L;

  const int rows = 64 * NB;
  const int nmine = a.bcount[d.wave];  // this wave's blocks: a.blist[wave][0..nmine)
  // K -> bf16x3 chunks, V -> V^T planes: one pass, all global loads of an iteration first
  for (int m0 = 0; m0 < rows; m0 += 2 * blockDim.x) {
    float kx[2][4], vx[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      load_row4(kx[u], kp, L, m);
      load_row4(vx[u], vp, L, m);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      if (m < rows) {
        put_chunks(kc0, kc2, m, kx[u]);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) vt[dd * Lp + m] = vx[u][dd];
      }
    }
  }
  if (threadIdx.x == 0) *ones = ones_chunk();
  __syncthreads();
  if (nmine == 0) return;
  // streamed-operand address of this lane: groups 0,1 -> c0[key], 2 -> c2[key], 3 -> the ones chunk
  const bf16x8* abase = g == 3 ? ones : (g == 2 ? kc2 : kc0) + qi;
  const int astride = g == 3 ? 0 : 1;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < nmine; ++it) {
    const int blk = a.blist[d.wave][it];
    const int q0 = 64 * blk;
    const int ngrp = min(4, (L - q0 + 15) >> 4);  // 16-query groups of this block that exist

    float mcur[4], lsum[4];
    bf16x8 bq[4];
    f32x4 acc[4];
    int qidx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qidx[t] = q0 + 16 * t + qi;
      float qv[4];
      load_row4(qv, qp, L, qidx[t]);
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) qv[dd] *= a.scale2;
      mcur[t] = 0.f;  // provisional: the first tile (FORCE) moves every query's m to its exact maximum
      bq[t] = resident_operand(qv, 0.f, g);
      lsum[t] = 0.f;
      acc[t] = zero4;
    }

    // move group t's running max (one per query, shared by its four key subsets) by c and rescale
    auto shift_max = [&](int t, float c) {
      const float alpha = ex2(-c);
      float al[4];
      quad_all(alpha, al);
      lsum[t] *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] *= al[i];
      mcur[t] += c;
      set_constant(bq[t], -mcur[t], g);
    };

    // One 16-key tile against groups [TMIN, 4): scores for all of them, exponentials, ONE overflow
    // test, then the P.V accumulation — 4 independent dependency chains per phase.
    //   MASK 0: no lane is cut (tile strictly below every diagonal)
    //   MASK 1: group TMIN is on its diagonal, per-lane causal predicate for it
    //   MASK 2: predicate for every group (the very first tile of block 0)
    //   FORCE : set every query's m to the exact maximum of its allowed scores (first tile)
    auto step = [&](int k0, const bf16x8& kf, const f32x4& vf, auto TMIN_, auto MASK_, auto FORCE_) {
      constexpr int TMIN = decltype(TMIN_)::value, MASK = decltype(MASK_)::value;
      constexpr bool FORCE = decltype(FORCE_)::value;
      f32x4 s[4];
      float p[4][4], ps[4];
      int lim[4];
#pragma unroll
      for (int t = TMIN; t < 4; ++t) s[t] = MFMA16B(kf, bq[t], zero4);
#pragma unroll
      for (int t = TMIN; t < 4; ++t) {
        const bool cut = MASK == 2 || (MASK == 1 && t == TMIN);
        lim[t] = cut ? qidx[t] - a.strict - (k0 + 4 * g) : 3;  // key 4g+r allowed iff r <= lim
#pragma unroll
        for (int r = 0; r < 4; ++r) p[t][r] = (!cut || r <= lim[t]) ? ex2(s[t][r]) : 0.f;
        ps[t] = (p[t][0] + p[t][1]) + (p[t][2] + p[t][3]);
      }
      float pmax = ps[TMIN];
#pragma unroll
      for (int t = TMIN + 1; t < 4; ++t) pmax = fmaxf(pmax, ps[t]);
      if (FORCE || __any(pmax > PSUM_TH)) {
        // some score ran more than ~2^8 above its query's running max: move the max up, rescale
#pragma unroll
        for (int t = TMIN; t < 4; ++t) {
          const bool cut = MASK == 2 || (MASK == 1 && t == TMIN);
          float c = NEG_BIG;
#pragma unroll
          for (int r = 0; r < 4; ++r) c = fmaxf(c, (!cut || r <= lim[t]) ? s[t][r] : NEG_BIG);
          c = gmax(c);
          c = c > 0.5f * NEG_BIG ? c : 0.f;  // query without any allowed key so far: leave m alone
          if (!FORCE) c = fmaxf(c, 0.f);
          shift_max(t, c);
#pragma unroll
          for (int r = 0; r < 4; ++r) p[t][r] = (!cut || r <= lim[t]) ? ex2(s[t][r] - c) : 0.f;
          ps[t] = (p[t][0] + p[t][1]) + (p[t][2] + p[t][3]);
        }
      }
#pragma unroll
      for (int t = TMIN; t < 4; ++t) lsum[t] += ps[t];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int t = TMIN; t < 4; ++t) acc[t] = MFMA4(p[t][r], vf[r], acc[t]);
      }
    };
    auto frag_k = [&](int k0) { return abase[astride * k0]; };
    auto frag_v = [&](int k0) { return *reinterpret_cast<const f32x4*>(vt + jc * Lp + k0 + 4 * g); };

    // ---- first tile: every query that will ever see an allowed key sees one here (queries >= 16:
    //      the whole tile; queries < 16 have no other tile)
    if (q0 == 0) step(0, frag_k(0), frag_v(0), I<0>{}, I<2>{}, B<true>{});
    else step(0, frag_k(0), frag_v(0), I<0>{}, I<0>{}, B<true>{});

    // ---- full tiles (strictly below every group's diagonal): 4 groups per K/V fragment
    bf16x8 kfn = frag_k(16);
    f32x4 vfn = frag_v(16);
    for (int k0 = 16; k0 < q0; k0 += 16) {
      const bf16x8 kf = kfn;
      const f32x4 vf = vfn;
      kfn = frag_k(k0 + 16);
      vfn = frag_v(k0 + 16);
      step(k0, kf, vf, I<0>{}, I<0>{}, B<false>{});
    }

    // ---- the block's own 64 keys: tile u meets groups t >= u, group u on its diagonal. (Groups
    //      beyond L in the last block run along on zero queries; they are never stored.)
    if (q0 != 0) step(q0, frag_k(q0), frag_v(q0), I<0>{}, I<1>{}, B<false>{});
    if (ngrp > 1) step(q0 + 16, frag_k(q0 + 16), frag_v(q0 + 16), I<1>{}, I<1>{}, B<false>{});
    if (ngrp > 2) step(q0 + 32, frag_k(q0 + 32), frag_v(q0 + 32), I<2>{}, I<1>{}, B<false>{});
    if (ngrp > 3) step(q0 + 48, frag_k(q0 + 48), frag_v(q0 + 48), I<3>{}, I<1>{}, B<false>{});

    // ---- sum the four key subsets g of every query (they share m), normalise, store
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ngrp) {
        const float l = gsum(lsum[t]);
        float iq[4], ov[4];
        quad_all(l > 0.f ? 1.f / l : 0.f, iq);
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = gsum(acc[t][i]) * iq[i];
        if (g == 0) {
          store_rows4(op + (size_t)jc * L, L, q0 + 16 * t + d.qb4, a.vec, ov[0], ov[1], ov[2], ov[3]);
          if (qidx[t] < L) lsep[qidx[t]] = l > 0.f ? mcur[t] + log2f(l) : POS_BIG;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ backward: dQ
// Also writes delta[l] = sum_j dO[l][j] * O[l][j] (read by the dK/dV kernel).
// LDS per workgroup: K chunks, V chunks (2 x 16 B per key each), K^T planes [4][Lp] fp32, ones chunk.
__global__ void __launch_bounds__(512) attn_dq_m44_kernel(const PgAttnArgs a) {
  extern __shared__ float4 lds4[];
  const int Lp = a.lp;
  bf16x8* kc0 = reinterpret_cast<bf16x8*>(lds4);
  bf16x8* kc2 = kc0 + Lp;
  bf16x8* vc0 = kc2 + Lp;
  bf16x8* vc2 = vc0 + Lp;
  float* kt = reinterpret_cast<float*>(vc2 + Lp);  // K^T [4][Lp]
  bf16x8* ones = reinterpret_cast<bf16x8*>(kt + 4 * Lp);
  const Ids d = ids();
  const int qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * 4 * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * 4 * L;
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * 4 * L;
  const size_t row = ((size_t)n * a.heads + h) * L;

  const int rows = 64 * NB;
  const int nmine = a.bcount[d.wave];  // this wave's blocks: a.blist[wave][0..nmine)
  // K, V -> bf16x3 chunks, K -> K^T planes: one pass, all global loads of an iteration first
  for (int m0 = 0; m0 < rows; m0 += 2 * blockDim.x) {
    float kx[2][4], vx[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      load_row4(kx[u], kp, L, m);
      load_row4(vx[u], vp, L, m);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      if (m < rows) {
        put_chunks(kc0, kc2, m, kx[u]);
        put_chunks(vc0, vc2, m, vx[u]);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) kt[dd * Lp + m] = kx[u][dd];
      }
    }
  }
  if (threadIdx.x == 0) *ones = ones_chunk();
  __syncthreads();
  if (nmine == 0) return;
  // streamed-operand addresses of this lane: groups 0,1 -> c0[key], 2 -> c2[key], 3 -> ones chunk
  const bf16x8* kbase = g == 3 ? ones : (g == 2 ? kc2 : kc0) + qi;
  const bf16x8* vbase = g == 3 ? ones : (g == 2 ? vc2 : vc0) + qi;
  const int astride = g == 3 ? 0 : 1;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < nmine; ++it) {
    const int blk = a.blist[d.wave][it];
    const int q0 = 64 * blk;
    const int ngrp = min(4, (L - q0 + 15) >> 4);

    bf16x8 bq[4], bg[4];  // resident operands: q (constant -lse2) and dO (constant -delta)
    f32x4 acc[4];
    int qidx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qidx[t] = q0 + 16 * t + qi;
      const bool ok = qidx[t] < L;
      float qv[4], gv[4], ov[4], dl = 0.f;
      load_row4(qv, qp, L, qidx[t]);
      load_row4(gv, gp, L, qidx[t]);
      load_row4(ov, op, L, qidx[t]);
      const float lse = ok ? a.lse2_in[row + qidx[t]] : POS_BIG;
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        qv[dd] *= a.scale2;
        dl = fmaf(gv[dd], ov[dd], dl);
      }
      if (ok && g == 0) a.delta[row + qidx[t]] = dl;
      bq[t] = resident_operand(qv, -lse, g);
      bg[t] = resident_operand(gv, -dl, g);
      acc[t] = zero4;
    }

    // One 16-key tile against groups [TMIN, 4): P = exp2(S - lse), dS = P * (dP - delta), dQ += dS K.
    // MASK: group TMIN is on its diagonal (per-lane causal predicate).
    struct Frag { bf16x8 ka, va; f32x4 kq; };
    auto frag = [&](int k0) {
      Frag f;
      f.ka = kbase[astride * k0];
      f.va = vbase[astride * k0];
      f.kq = *reinterpret_cast<const f32x4*>(kt + jc * Lp + k0 + 4 * g);
      return f;
    };
    auto step = [&](int k0, const Frag& f, auto TMIN_, auto MASK_) {
      constexpr int TMIN = decltype(TMIN_)::value;
      constexpr bool MASK = decltype(MASK_)::value;
      f32x4 s[4], dp[4];
      float ds[4][4];
#pragma unroll
      for (int t = TMIN; t < 4; ++t) {
        s[t] = MFMA16B(f.ka, bq[t], zero4);
        dp[t] = MFMA16B(f.va, bg[t], zero4);
      }
#pragma unroll
      for (int t = TMIN; t < 4; ++t) {
        const bool cut = MASK && t == TMIN;
        const int lim = cut ? qidx[t] - a.strict - (k0 + 4 * g) : 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[t][r] = (!cut || r <= lim) ? ex2(s[t][r]) * dp[t][r] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int t = TMIN; t < 4; ++t) acc[t] = MFMA4(ds[t][r], f.kq[r], acc[t]);
      }
    };
    // full tiles
    Frag fn = frag(0);
    for (int k0 = 0; k0 < q0; k0 += 16) {
      const Frag f = fn;
      fn = frag(k0 + 16);
      step(k0, f, I<0>{}, B<false>{});
    }
    // the block's own 64 keys: tile u meets groups t >= u, group u on its diagonal
    step(q0, fn, I<0>{}, B<true>{});
    if (ngrp > 1) step(q0 + 16, frag(q0 + 16), I<1>{}, B<true>{});
    if (ngrp > 2) step(q0 + 32, frag(q0 + 32), I<2>{}, B<true>{});
    if (ngrp > 3) step(q0 + 48, frag(q0 + 48), I<3>{}, B<true>{});
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ngrp) {
        float o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o4[i] = gsum(acc[t][i]) * a.scale;
        if (g == 0) store_rows4(dqp + (size_t)jc * L, L, q0 + 16 * t + d.qb4, a.vec, o4[0], o4[1], o4[2], o4[3]);
      }
    }
  }
}

// ------------------------------------------------------------------------------ backward: dK, dV
// Owner = 64-key block (four 16-key groups); the queries stream. Tile D layout: lane (key j, g),
// VGPR r <-> query 4g+r.
// BF16S: the score tile S = Q K^T runs on the bf16x3 MFMA (q chunks streamed from LDS, +32 B per query:
// 61 KB per workgroup, so 8 waves x 2 workgroups per CU instead of 4 x 4); dP stays on the fp32 tile.
template <bool BF16S>
__global__ void __launch_bounds__(512) attn_dkv_m44_kernel(const PgAttnArgs a) {
  extern __shared__ float4 lds4[];
  const int Lp = a.lp;
  bf16x8* qc0 = reinterpret_cast<bf16x8*>(lds4);  // q chunks [yh|ym], [yl|yh] (BF16S only)
  bf16x8* qc2 = qc0 + Lp;
  float* qt = reinterpret_cast<float*>(lds4) + (BF16S ? 8 * Lp : 0);  // Q^T  [4][Lp]
  float* gt = qt + 4 * Lp;                     // dO^T [4][Lp]
  float* nlse = gt + 4 * Lp;                   // -lse2 [Lp]   (-BIG for rows >= L: P = 0)
  float* ndel = nlse + Lp;                     // -delta [Lp]
  const Ids d = ids();
  const int qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * 4 * L;
  float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * 4 * L;
  float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * 4 * L;
  const size_t row = ((size_t)n * a.heads + h) * L;

  const int r0 = 0, r1 = 64 * NB;
  const int nmine = a.bcount[d.wave];  // this wave's key blocks: a.blist[wave][0..nmine)
  // q, dO -> planes, -lse2, -delta: one pass, all global loads of an iteration first
  for (int m0 = r0; m0 < r1; m0 += 2 * blockDim.x) {
    float qx[2][4], gx[2][4], lx[2], dx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      load_row4(qx[u], qp, L, m);
      load_row4(gx[u], gp, L, m);
      const int mc = m < L ? m : L - 1;
      lx[u] = a.lse2_in[row + mc];
      dx[u] = a.delta[row + mc];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      if (m < r1) {
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          qt[dd * Lp + m] = qx[u][dd];
          gt[dd * Lp + m] = gx[u][dd];
        }
        if (BF16S) put_chunks(qc0, qc2, m, qx[u]);
        nlse[m] = m < L ? -lx[u] : NEG_BIG;
        ndel[m] = m < L ? -dx[u] : 0.f;
      }
    }
  }
  __syncthreads();
  if (nmine == 0) return;
  const int q_end = ((L + 15) >> 4) << 4;
  // streamed score operand of this lane: groups 0,1 -> c0[query], 2 -> c2[query]; group 3 pairs with
  // a zero resident operand (the constants -lse2 ride in the C operand here), any chunk will do
  const bf16x8* qbase = (g == 2 ? qc2 : qc0) + qi;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < nmine; ++it) {
    const int blk = a.blist[d.wave][it];
    const int kb0 = 64 * blk;
    const int ngrp = min(4, (L - kb0 + 15) >> 4);

    float kf[4], vf[4];
    bf16x8 bk[4];
    f32x4 acck[4], accv[4];
    int kidx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kidx[t] = kb0 + 16 * t + qi;
      const bool ok = kidx[t] < L;
      const int kc = ok ? kidx[t] : L - 1;
      kf[t] = ok ? kp[(size_t)g * L + kc] * a.scale2 : 0.f;
      vf[t] = ok ? vp[(size_t)g * L + kc] : 0.f;
      if (BF16S) {
        float k4[4];
        load_row4(k4, kp, L, kidx[t]);
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) k4[dd] *= a.scale2;
        bk[t] = resident_operand(k4, 0.f, g);
      }
      acck[t] = zero4;
      accv[t] = zero4;
    }

    // One 16-query tile against key groups [0, TMAX]: P = exp2(S - lse), dS = P * (dP - delta),
    // dV += P^T dO, dK += dS^T Q. MASK: group TMAX is on its diagonal (per-lane causal predicate).
    struct Frag { float qa, ga; bf16x8 qa8; f32x4 cl, cd, qq, gq; };
    auto frag = [&](int q0t) {
      Frag f;
      if (BF16S) f.qa8 = qbase[q0t];
      else f.qa = qt[g * Lp + q0t + qi];
      f.ga = gt[g * Lp + q0t + qi];
      f.cl = *reinterpret_cast<const f32x4*>(nlse + q0t + 4 * g);
      f.cd = *reinterpret_cast<const f32x4*>(ndel + q0t + 4 * g);
      f.qq = *reinterpret_cast<const f32x4*>(qt + jc * Lp + q0t + 4 * g);
      f.gq = *reinterpret_cast<const f32x4*>(gt + jc * Lp + q0t + 4 * g);
      return f;
    };
    auto step = [&](int q0t, const Frag& f, auto TMAX_, auto MASK_) {
      constexpr int TMAX = decltype(TMAX_)::value;
      constexpr bool MASK = decltype(MASK_)::value;
      f32x4 s[4], dp[4];
      float p[4][4], ds[4][4];
#pragma unroll
      for (int t = 0; t <= TMAX; ++t) {
        if (BF16S) s[t] = MFMA16B(f.qa8, bk[t], f.cl);
        else s[t] = MFMA16(f.qa, kf[t], f.cl);
        dp[t] = MFMA16(f.ga, vf[t], f.cd);
      }
#pragma unroll
      for (int t = 0; t <= TMAX; ++t) {
        const bool cut = MASK && t == TMAX;
        const int lowest = cut ? kidx[t] + a.strict - (q0t + 4 * g) : 0;  // query 4g+r allowed iff r >= lowest
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[t][r] = (!cut || r >= lowest) ? ex2(s[t][r]) : 0.f;
          ds[t][r] = p[t][r] * dp[t][r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int t = 0; t <= TMAX; ++t) {
          accv[t] = MFMA4(p[t][r], f.gq[r], accv[t]);
          acck[t] = MFMA4(ds[t][r], f.qq[r], acck[t]);
        }
      }
    };
    // the block's own 64 queries: tile u meets key groups t <= u, group u on its diagonal
    step(kb0, frag(kb0), I<0>{}, B<true>{});
    if (ngrp > 1) step(kb0 + 16, frag(kb0 + 16), I<1>{}, B<true>{});
    if (ngrp > 2) step(kb0 + 32, frag(kb0 + 32), I<2>{}, B<true>{});
    if (ngrp > 3) step(kb0 + 48, frag(kb0 + 48), I<3>{}, B<true>{});
    // every later query tile: all four key groups, no masks (rows up to 64*NB + 15 are inside the
    // planes, so the fragment prefetch may run one tile past the end)
    if (kb0 + 64 < q_end) {
      Frag fn = frag(kb0 + 64);
      for (int q0t = kb0 + 64; q0t < q_end; q0t += 16) {
        const Frag f = fn;
        fn = frag(q0t + 16);
        step(q0t, f, I<3>{}, B<false>{});
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ngrp) {
        float k4[4], v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          k4[i] = gsum(acck[t][i]) * a.scale;
          v4[i] = gsum(accv[t][i]);
        }
        if (g == 0) {
          store_rows4(dkp + (size_t)jc * L, L, kb0 + 16 * t + d.qb4, a.vec, k4[0], k4[1], k4[2], k4[3]);
          store_rows4(dvp + (size_t)jc * L, L, kb0 + 16 * t + d.qb4, a.vec, v4[0], v4[1], v4[2], v4[3]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ backward, fused
// dQ, dK and dV in ONE pass over the allowed (query, key) pairs (round 3). The two-kernel backward above
// evaluates S, dP and exp2 twice (once per orientation: 24 + 32 = 56 FLOP and 2 exp per pair); here a
// pair costs 40 FLOP and 1 exp. Orientation = the dK/dV kernel's (owner = 64-key block, queries
// stream, tile D layout lane (key j, g), VGPR r <-> query 4g+r), so dV += P^T dO and dK += dS^T Q
// accumulate in registers as before. dQ += dS K contracts over KEYS, which live on the lane axis of
// that layout: each 16x16 dS tile is transposed through a per-wave 1 KB LDS scratch (one
// ds_write_b128 + four ds_read_b32), then the transposed tile's registers are the A operands of four
// more 4x4x1 MFMAs against resident K^T values (the partial products of a wave's four key groups for
// one 16-query tile add up in ONE accumulator). Per 16-query x 64-key step that accumulator is summed
// over the four lane groups through the same scratch (one ds_write_b128, four ds_read_b32: lane l then
// owns output l = (channel, query)) and deposited into a per-workgroup dQ plane in LDS; the planes
// are written to HBM once at the end. The deposit is NOT ds_add_f32: measured on MI355X
// (tools/exp/lds_atomic_ubench.hip) an LDS fp32 atomic add costs 194 cycles per wave-instruction per CU
// whatever its addresses (ds_add_u32: 4.7, a plain read + write: 11) — with four of them per step the
// first version of this kernel ran 2.05 ms against 0.99 ms for the two kernels it replaces. Instead a
// lane takes the slot's content with an integer exchange (old = xchg(slot, 0); x += old), puts the sum
// back (old2 = xchg(slot, x)) and, if another wave deposited in between (old2 != 0), carries old2 into
// another round: nothing is ever lost, no lock, two integer-speed atomics in the common case.
// (dQ's summation order over key blocks varies from run to run in the last bit; dK / dV stay
// bit-reproducible. PG_ATTN_FUSED_BWD=0 selects the two-kernel path.)
//   scratch layout (both directions conflict free under the per-instruction banking of
//   MI355X_MICROARCH.md §LDS: ds_write_b128 = 8-lane groups, ds_read_b32 = 32-lane groups, 32 banks):
//   element (key k, query q) of the 16 x 16 tile lives in 16-byte slot 16 G + ((k + 2 G) & 15),
//   G = q >> 2, at dword q & 3. A writing lane (key k, lane group G) stores its four queries as one
//   slot; a reading lane (query q, lane group g') takes keys g' + 4 s, s = 0..3 — for one s the 32
//   lanes of a read group cover slots (g' + 4 s + 2 G) & 7 = all eight residues — and the resident
//   K^T values are laid out in that key order.
// S runs on the bf16x3 MFMA with -lse2 folded into the contraction (32-byte query rows
// [qh qm | ql c_h c_m c_l 0] against resident key operands [kh kh], [km km], [kh 1 1 1 0], [kl 0]
// for lane groups 0..3: qh.kh + qm.kh + qh.km + qm.km + ql.kh + qh.kl + c), dP on the fp32 16x16x4
// tile with -delta in the C operand; delta = sum_j dO O is computed while staging (no delta tensor).
// LDS: 21 planes of Lp floats + 8 KB scratch = 79.4 KB at L = 784: two 8-wave workgroups per CU.
// Measured on MI355X at N = 1024, 4 heads, L = 784 (tools/exp/attn_bwd_ab.py, gpurun_out r3c): the two
// kernels above 465 + 544 = 1010 us, this kernel 752 us (all results within 1e-6 of theirs). Ablation
// (runtime switches since removed: they split the step into basic blocks): without the deposit 723,
// also without the transposition 701, also without the dQ MFMAs 629. Variants built and measured, then removed: key groups all four at a time 953 us and
// register prefetch of the next tile's fragments 1092 us (both spill at 128 registers); 6-wave
// workgroups with up to 168 registers (3 waves per SIMD, no spills) 1257-1266 us.
__device__ __forceinline__ bf16x8 resident_operand_k(const float (&x)[4], int kg) {
  bf16x8 v;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const Split3 s = split3(x[dd]);
    v[dd] = kg == 1 ? s.m : (kg == 3 ? s.l : s.h);
    v[4 + dd] = kg == 0 ? s.h : (kg == 1 ? s.m : (__bf16)0.f);
  }
  if (kg == 2) { v[4] = v[5] = v[6] = (__bf16)1.f; }
  return v;
}

__device__ __forceinline__ void put_row32(bf16x8* __restrict__ w1, bf16x8* __restrict__ w2, int row,
                                          const float (&y)[4], float c) {
  bf16x8 a, b;
#pragma unroll
  for (int dd = 0; dd < 4; ++dd) {
    const Split3 s = split3(y[dd]);
    a[dd] = s.h; a[4 + dd] = s.m;
    b[dd] = s.l;
  }
  const Split3 cc = split3(c);
  b[4] = cc.h; b[5] = cc.m; b[6] = cc.l; b[7] = (__bf16)0.f;
  w1[row] = a;
  w2[row] = b;
}

// LPC: the LDS plane stride as a compile-time constant (848 = 28 x 28 images, 1040 = 32 x 32; 0 = runtime
// a.lp): every plane pointer then is ONE lane base + an immediate offset instead of a register each
template <int LPC>
__global__ void __launch_bounds__(512, 4) attn_bwd_m44_kernel(const PgAttnArgs a) {
  constexpr int TG = 2;  // key groups processed together (see the variants measured below)
  extern __shared__ float4 lds4[];
  const int Lp = LPC > 0 ? LPC : a.lp;
  bf16x8* qw1 = reinterpret_cast<bf16x8*>(lds4);          // [Lp] [qh qm]      (16-byte stride: conflict-free
  bf16x8* qw2 = qw1 + Lp;                                 // [Lp] [ql c 0]      ds_read_b128 fragment reads)
  float* qt = reinterpret_cast<float*>(lds4) + 8 * Lp;    // Q^T  [4][Lp]
  float* gt = qt + 4 * Lp;                                // dO^T [4][Lp]
  float* ndel = gt + 4 * Lp;                              // -delta [Lp]
  float* dqa = ndel + Lp;                                 // dQ accumulators [4][Lp]
  const Ids d = ids();
  float* scr = dqa + 4 * Lp + d.wave * 256;               // this wave's 16 x 16 transposition scratch
  const int qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * 4 * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * 4 * L;
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * 4 * L;
  float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * 4 * L;
  float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * 4 * L;
  const size_t row = ((size_t)n * a.heads + h) * L;

  const int r1 = 64 * NB;
  const int nmine = a.bcount[d.wave];  // this wave's key blocks: a.blist[wave][0..nmine)
  // q, dO -> planes + 32-byte rows, delta, zeroed dQ planes: one pass, all global loads of an iteration first
  for (int m0 = 0; m0 < r1; m0 += 2 * blockDim.x) {
    float qx[2][4], gx[2][4], ox[2][4], lx[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      load_row4(qx[u], qp, L, m);
      load_row4(gx[u], gp, L, m);
      load_row4(ox[u], op, L, m);
      lx[u] = a.lse2_in[row + (m < L ? m : L - 1)];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m0 + u * blockDim.x + threadIdx.x;
      if (m < r1) {
        float dl = 0.f;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          qt[dd * Lp + m] = qx[u][dd];
          gt[dd * Lp + m] = gx[u][dd];
          dqa[dd * Lp + m] = 0.f;
          dl = fmaf(gx[u][dd], ox[u][dd], dl);
        }
        put_row32(qw1, qw2, m, qx[u], m < L ? -lx[u] : NEG_BIG);
        ndel[m] = -dl;
      }
    }
  }
  __syncthreads();
  const int q_end = ((L + 15) >> 4) << 4;
  // streamed score operand of this lane: lane groups 0, 1, 3 read [qh qm], group 2 reads [ql c]
  const bf16x8* qbase = (g == 2 ? qw2 : qw1) + qi;
  // transposition scratch: this lane writes its key's row (permuted), reads column qi of rows 4g..4g+3
  float* scr_w = scr + 4 * (16 * g + ((qi + 2 * g) & 15));
  const float* scr_r[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl)
    scr_r[sl] = scr + 4 * (16 * (qi >> 2) + ((g + 4 * sl + 2 * (qi >> 2)) & 15)) + (qi & 3);
  float* red_w = scr + 64 * g + 16 * jc + d.qb4;
  const float* red_r = scr + d.lane;
  unsigned int* dq_slot = reinterpret_cast<unsigned int*>(dqa + (d.lane >> 4) * Lp + (d.lane & 15));

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int it = 0; it < nmine; ++it) {
    const int blk = a.blist[d.wave][it];
    const int kb0 = 64 * blk;
    const int ngrp = min(4, (L - kb0 + 15) >> 4);

    float vf[4];
    bf16x8 bk[4];
    f32x4 kq[4];  // K^T values of the transposed tile's registers: kq[t][s] = K[kb0 + 16 t + g + 4 s][jc]
    f32x4 acck[4], accv[4];
    int kidx[4];
    // per-lane row selectors of the block set-up, opaque so that the 64-bit row pointers (v + g L, k + jc L) are
    // formed here, once per block, instead of living in registers across the block loop
    int g_row = g, jc_row = jc;
    asm volatile("" : "+v"(g_row), "+v"(jc_row));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kidx[t] = kb0 + 16 * t + qi;
      const bool ok = kidx[t] < L;
      vf[t] = ok ? vp[(size_t)g_row * L + kidx[t]] : 0.f;
      float k4[4];
      load_row4(k4, kp, L, kidx[t]);
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) k4[dd] *= a.scale2;
      bk[t] = resident_operand_k(k4, g);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const int kk = kb0 + 16 * t + g + 4 * sl;
        const float kv = kp[(size_t)jc_row * L + (kk < L ? kk : L - 1)];
        kq[t][sl] = kk < L ? kv : 0.f;
      }
      acck[t] = zero4;
      accv[t] = zero4;
    }

    // One 16-query tile against key groups [0, TMAX]: P = exp2(S - lse), dS = P * (dP - delta),
    // dV += P^T dO, dK += dS^T Q, dQ += dS K. MASK: group TMAX is on its diagonal.
    struct Frag { float ga; bf16x8 qa8; f32x4 cd, qq, gq; };
    auto frag = [&](int q0t) {
      Frag f;
      f.qa8 = qbase[q0t];
      f.ga = gt[g * Lp + q0t + qi];
      f.cd = *reinterpret_cast<const f32x4*>(ndel + q0t + 4 * g);
      f.qq = *reinterpret_cast<const f32x4*>(qt + jc * Lp + q0t + 4 * g);
      f.gq = *reinterpret_cast<const f32x4*>(gt + jc * Lp + q0t + 4 * g);
      return f;
    };
    auto step = [&](int q0t, const Frag& f, auto TMAX_, auto MASK_) {
      constexpr int TMAX = decltype(TMAX_)::value;
      constexpr bool MASK = decltype(MASK_)::value;
      f32x4 dqacc = zero4;
      // TG key groups at a time: their score / exp / product chains are independent, and the LDS round
      // trip of one tile's transposition runs under the dV / dK MFMAs of the next (a wave's LDS
      // operations execute in order: a tile's reads are queued before the next tile's write, so ONE
      // scratch tile per wave suffices and no wait sits between a write and its reads)
#pragma unroll
      for (int t0 = 0; t0 <= TMAX; t0 += TG) {
        f32x4 s[TG], dp[TG], ds[TG];
        float p[TG][4], dst[TG][4];
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int t = t0 + u;
          if (t <= TMAX) {
            s[u] = MFMA16B(f.qa8, bk[t], zero4);
            dp[u] = MFMA16(f.ga, vf[t], f.cd);
          }
        }
#pragma unroll
        for (int u = 0; u < TG; ++u) {
          const int t = t0 + u;
          if (t <= TMAX) {
            const bool cut = MASK && t == TMAX;
            const int lowest = cut ? kidx[t] + a.strict - (q0t + 4 * g) : 0;  // query 4g+r allowed iff r >= lowest
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              p[u][r] = (!cut || r >= lowest) ? ex2(s[u][r]) : 0.f;
              ds[u][r] = p[u][r] * dp[u][r];
            }
          }
        }
#pragma unroll
        for (int u = 0; u <= TG; ++u) {
          const int t = t0 + u;
          if (u < TG && t <= TMAX) {
            *reinterpret_cast<f32x4*>(scr_w) = ds[u];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) dst[u][sl] = *scr_r[sl];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              accv[t] = MFMA4(p[u][r], f.gq[r], accv[t]);
              acck[t] = MFMA4(ds[u][r], f.qq[r], acck[t]);
            }
          }
          if (u > 0 && t - 1 <= TMAX) {  // dQ of the previous tile: its transposed values have had a tile's time
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) dqacc = MFMA4(dst[u - 1][sl], kq[t - 1][sl], dqacc);
          }
        }
      }
      // D_b[i'][j'] of block (g, qb): query q0t + 4 qb + i', channel jc — partial over this lane group's keys
      {
        // sum over the four lane groups through the scratch (the last tile's transposed values have been
        // consumed: LDS operations of a wave execute in order), then lane l owns output (channel l >> 4,
        // query q0t + (l & 15))
        *reinterpret_cast<f32x4*>(red_w) = dqacc;
        asm volatile("" ::: "memory");
        float x = (red_r[0] + red_r[64]) + (red_r[128] + red_r[192]);
        asm volatile("" ::: "memory");
        unsigned int* slot = dq_slot + q0t;
        for (;;) {
          x += __uint_as_float(atomicExch(slot, 0u));
          const unsigned int back = atomicExch(slot, __float_as_uint(x));
          if (__uint_as_float(back) == 0.f) break;  // the slot was empty: deposited
          x = __uint_as_float(back);                // another wave's deposit came in between: carry it on
        }
      }
    };
    // the block's own 64 queries: tile u meets key groups t <= u, group u on its diagonal
    step(kb0, frag(kb0), I<0>{}, B<true>{});
    if (ngrp > 1) step(kb0 + 16, frag(kb0 + 16), I<1>{}, B<true>{});
    if (ngrp > 2) step(kb0 + 32, frag(kb0 + 32), I<2>{}, B<true>{});
    if (ngrp > 3) step(kb0 + 48, frag(kb0 + 48), I<3>{}, B<true>{});
    // every later query tile: all four key groups, no masks. (No register prefetch of the next tile's
    // fragments: at 128 registers it spills; the other three waves of the SIMD cover the LDS latency.)
    for (int q0t = kb0 + 64; q0t < q_end; q0t += 16) step(q0t, frag(q0t), I<3>{}, B<false>{});
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ngrp) {
        float k4[4], v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          k4[i] = gsum(acck[t][i]) * a.scale;
          v4[i] = gsum(accv[t][i]);
        }
        if (g == 0) {
          // the lane's channel row, opaque to the optimiser: with a visible jc it keeps the two per-lane 64-bit row
          // pointers alive across the whole block loop — at the kernel's 128 registers that was 6 spilled VGPRs
          int jcs = jc;
          asm volatile("" : "+v"(jcs));
          store_rows4(dkp + (size_t)jcs * L, L, kb0 + 16 * t + d.qb4, a.vec, k4[0], k4[1], k4[2], k4[3]);
          store_rows4(dvp + (size_t)jcs * L, L, kb0 + 16 * t + d.qb4, a.vec, v4[0], v4[1], v4[2], v4[3]);
        }
      }
    }
  }
  __syncthreads();  // every wave's dQ contributions are in the planes
#pragma unroll
  for (int dd = 0; dd < 4; ++dd)
    for (int m = threadIdx.x; m < L; m += blockDim.x) dqp[(size_t)dd * L + m] = dqa[dd * Lp + m] * a.scale;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// waves per workgroup: forward / dK,dV use 4 (a multiple of the 4 SIMDs; 3-4 workgroups per CU),
// dQ (68 KB of LDS: two workgroups per CU) 8. PG_ATTN_WAVES="f,q,k" overrides for tuning.
struct AttnWaveCfg { int w[3]; };
static void attn_waves(int* w) {
  // function-local static with an initialiser: thread-safe one-time init (the library is entered from
  // the main thread and from the autograd thread)
  static const AttnWaveCfg cfg = []() {
    AttnWaveCfg c = {{4, 8, 4}};
    if (const char* e = PG_AB_ENV("PG_ATTN_WAVES")) {
      int f = 0, q = 0, k = 0;
      if (sscanf(e, "%d,%d,%d", &f, &q, &k) == 3 && f >= 1 && f <= 8 && q >= 1 && q <= 8 && k >= 1 && k <= 8)
        c = {{f, q, k}};
    }
    return c;
  }();
  w[0] = cfg.w[0]; w[1] = cfg.w[1]; w[2] = cfg.w[2];
}

int pg_attn_k4_launch(int which, const PgAttnArgs& a, hipStream_t st);  // attention_k4.hip

// process-wide switch of the fused backward (default: on, or PG_ATTN_FUSED_BWD=0/1 at load time);
// pg_attn_fused_bwd(0) selects the two-kernel backward, whose results are bit-reproducible
#include <atomic>
static std::atomic<int>& fused_bwd_flag() {
  static std::atomic<int> flag([]() { const char* e = PG_AB_ENV("PG_ATTN_FUSED_BWD"); return (e && e[0] == '0') ? 0 : 1; }());
  return flag;
}
static bool pg_attn_fused_bwd_enabled() { return fused_bwd_flag().load(std::memory_order_relaxed) != 0; }
PG_EXPORT int pg_attn_fused_bwd(int enable) {
  if (enable < 0) return fused_bwd_flag().load(std::memory_order_relaxed);
  return fused_bwd_flag().exchange(enable ? 1 : 0, std::memory_order_relaxed);
}

int pg_attn_mfma_launch(int which, const PgAttnArgs& a0, hipStream_t st) {
  if (which == PG_ATTN_BWD) {
    // fused backward: d_k = d_v = 4 only; PG_ATTN_FUSED_BWD=0 keeps the two-kernel backward (A/B, and
    // bit-reproducible dQ: the fused kernel sums dQ over key blocks with fp32 LDS atomics)
    if (!pg_attn_fused_bwd_enabled()) return 0;
    if (a0.dk_dim != 4 || a0.dv_dim != 4) {
      // d_k = 4, d_v = 16 / 32 (PixelSNAIL): attn_bwd_k4_kernel (round 4); PG_ATTN_FUSED_BWD_K4=0 for A/B
      static const bool k4_fused = []() { const char* e = PG_AB_ENV("PG_ATTN_FUSED_BWD_K4"); return !(e && e[0] == '0'); }();
      return k4_fused ? pg_attn_k4_launch(which, a0, st) : 0;
    }
  }
  // everything but d_k = d_v = 4: attention_k4.hip decides (d_k in {4, 16, 32, 64} x d_v in {16, 32, 64}, L % 16 == 0)
  if (a0.dk_dim != 4 || a0.dv_dim != 4) return pg_attn_k4_launch(which, a0, st);
  PgAttnArgs a = a0;
  const int NB = (a.L + 63) / 64;
  a.lp = 64 * NB + 16;  // plane stride == 16 (mod 64): conflict-free b32 and b128 fragment reads
  // in 4-byte units per row — fwd: K chunks (2 x 16 B) + V^T planes; dQ: K and V chunks + K^T planes;
  // dK/dV: 10 planes; plus the ones chunk
  // dK/dV score tile on the bf16x3 MFMA (default; PG_ATTN_DKV_BF16=0 selects the all-fp32 variant:
  // measured 0.534-0.545 ms vs 0.515-0.521 ms per launch at batch 1024)
  static const bool dkv_bf16 = []() { const char* e = PG_AB_ENV("PG_ATTN_DKV_BF16"); return !(e && e[0] == '0'); }();
  const size_t planes = which == PG_ATTN_BWD ? 21
                        : which == PG_ATTN_DKV ? (dkv_bf16 ? 18 : 10) : (which == PG_ATTN_DQ ? 20 : 12);
  int wcfg[3];
  attn_waves(wcfg);
  static const int bwd_waves = []() {
    const char* e = PG_AB_ENV("PG_ATTN_BWD_WAVES");
    const int v = e ? atoi(e) : 8;
    return v >= 1 && v <= 8 ? v : 8;
  }();
  int W = which == PG_ATTN_BWD ? bwd_waves : wcfg[which];
  // fused backward: + 1 KB of transposition scratch per wave
  const size_t shmem = planes * (size_t)a.lp * sizeof(float) + (which == PG_ATTN_BWD ? 8 * 1024 : 16);
  if (shmem > 160 * 1024) return 0;
  if (which == PG_ATTN_DKV && dkv_bf16 && !PG_AB_ENV("PG_ATTN_WAVES")) W = 8;
  // few (n, head) units (the reference's default batch 64 x 4 heads = one workgroup per CU): the launch
  // lasts as long as its most loaded wave, so the forward kernel also spreads its 13 query blocks over
  // 8 waves (longest list 94 -> 58 cost units; with many units per CU 4-wave workgroups pack better)
  if (which == PG_ATTN_FWD && (long)a.N * a.heads <= 512 && !PG_AB_ENV("PG_ATTN_WAVES")) W = 8;
  if (W > NB) W = NB;
  if ((NB + W - 1) / W > 16) return 0;  // block lists hold 16 entries per wave
  // LPT: blocks by decreasing cost (later query blocks / earlier key blocks stream more), each to
  // the least loaded wave
  long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int w = 0; w < 8; ++w) a.bcount[w] = 0;
  for (int rank = NB - 1; rank >= 0; --rank) {
    const int blk = (which == PG_ATTN_DKV || which == PG_ATTN_BWD) ? NB - 1 - rank : rank;
    int best = 0;
    for (int w = 1; w < W; ++w)
      if (load[w] < load[best] && a.bcount[w] < 16) best = w;
    if (a.bcount[best] >= 16) return 0;
    a.blist[best][a.bcount[best]++] = (unsigned char)blk;
    load[best] += 4 * rank + 5;
  }
  a.vec = a.L % 4 == 0 ? 1 : 0;
  if (which == PG_ATTN_FWD)
    a.vec = a.vec && aligned16(a.o_out) && a.o_bs % 4 == 0;
  else if (which == PG_ATTN_DQ)
    a.vec = a.vec && aligned16(a.dq) && a.dq_bs % 4 == 0;
  else  // dK/dV and the fused backward (whose dQ planes are written with scalar stores)
    a.vec = a.vec && aligned16(a.dk) && aligned16(a.dv) && a.dk_bs % 4 == 0 && a.dv_bs % 4 == 0;
  dim3 grid(1u, (unsigned)a.heads, (unsigned)a.N);
  dim3 block((unsigned)(64 * W));
  const void* fn = which == PG_ATTN_FWD  ? reinterpret_cast<const void*>(attn_fwd_m44_kernel)
                   : which == PG_ATTN_BWD ? (a.lp == 848 ? reinterpret_cast<const void*>(attn_bwd_m44_kernel<848>)
                                             : a.lp == 1040 ? reinterpret_cast<const void*>(attn_bwd_m44_kernel<1040>)
                                                            : reinterpret_cast<const void*>(attn_bwd_m44_kernel<0>))
                   : which == PG_ATTN_DQ ? reinterpret_cast<const void*>(attn_dq_m44_kernel)
                   : dkv_bf16            ? reinterpret_cast<const void*>(attn_dkv_m44_kernel<true>)
                                         : reinterpret_cast<const void*>(attn_dkv_m44_kernel<false>);
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (which == PG_ATTN_FWD)
    hipLaunchKernelGGL(attn_fwd_m44_kernel, grid, block, shmem, st, a);
  else if (which == PG_ATTN_BWD && a.lp == 848)
    hipLaunchKernelGGL(attn_bwd_m44_kernel<848>, grid, block, shmem, st, a);
  else if (which == PG_ATTN_BWD && a.lp == 1040)
    hipLaunchKernelGGL(attn_bwd_m44_kernel<1040>, grid, block, shmem, st, a);
  else if (which == PG_ATTN_BWD)
    hipLaunchKernelGGL(attn_bwd_m44_kernel<0>, grid, block, shmem, st, a);
  else if (which == PG_ATTN_DQ)
    hipLaunchKernelGGL(attn_dq_m44_kernel, grid, block, shmem, st, a);
  else if (dkv_bf16)
    hipLaunchKernelGGL(attn_dkv_m44_kernel<true>, grid, block, shmem, st, a);
  else
    hipLaunchKernelGGL(attn_dkv_m44_kernel<false>, grid, block, shmem, st, a);
  return 1;
}
