// attention_mfma.hip — matrix-core causal attention (forward, dQ, dK/dV) for d_k = d_v = 4, gfx950.
//
// ImageGPT (16 embed / 4 heads, BASELINE.json configs[1]) has d_k = d_v = 4: exactly the
// contraction depth of v_mfma_f32_16x16x4_f32 and exactly the tile edge of v_mfma_f32_4x4x1_16b_f32.
// Both are fp32 in / fp32 accumulate — no precision is given up. Measured on MI355X
// (tools/exp/ubench.hip): an fp32 MFMA and VALU work do NOT overlap on a SIMD
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0; MFMA + VALU streams cost the sum of their parts), so the matrix
// pipe buys cheaper multiply-adds, not a second pipe: a 16x16x4 tile costs 33 cycles for 4 scores
// per lane (16 v_fma = 58), a 4x4x1 costs 10.5 for 4 output FMAs per lane (14.4), v_exp_f32 10-12.
//
// One 16 keys x 16 queries tile, forward:
//   S^T = K Q^T            v_mfma_f32_16x16x4_f32   A[i][k] = K^T[k][key0+i]   (ds_read_b32)
//                                                   B[k][j] = q[query0+j][k]   (VGPR, held)
//                                                   C       = -m (running max, splat) -> D = s - m
//     D layout: lane (j = lane&15, g = lane>>4), VGPR r  <->  key 4g+r, query j
//   P = exp2(S^T)          4 x v_exp_f32 per lane
//   O += P^T V             4 x v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products):
//     block b = lane>>2 = (g, query quad qb); step r: A_b[i'] = P[key 4g+r][query 4qb+i'] — VGPR r
//     of the S^T tile AS IT IS, no cross-lane movement; B_b[j'] = V[key 4g+r][j'] — component r of
//     one ds_read_b128 from the V^T plane; D_b[i'][j'] = O partial [query 4qb+i'][channel j'].
//   No lane is padding in either instruction. The four key subsets g of a query keep lane-local
//   running max / sum / output and are merged once per 64-query block.
// The running max is folded into the C operand (s - m costs nothing) and updated LAZILY: the first
// tile sets every lane's m to its exact maximum; afterwards m moves only when a tile's probability
// sum shows that some score ran more than ~2^8 above it (one wave-uniform test per 4 groups).
// dQ and dK/dV use the same two instructions: S and dP^T = V dO^T (C = -lse2 resp. -delta, so
// exp2(D) = P and D = dP - delta come straight out of the matrix pipe), then dQ += dS K, or
// dV += P^T dO and dK += dS^T Q as 4x4x1 outer-product accumulations.
// A wave works on a 64-row block = four 16-row groups at a time (one K/V fragment feeds four
// independent MFMA/exp chains); blocks are handed out in balanced pairs exactly as in attention.hip.
// Measured (N=1024, 4 heads, L=784): fwd 0.45 ms, dQ 0.56 ms, dK/dV 0.62 ms vs 0.62 / 0.65 / 0.69 ms
// for the VALU row-owner kernels.
#include <type_traits>

#include "attention_args.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float NEG_BIG = -1.0e30f;
constexpr float POS_BIG = 1.0e30f;
constexpr float PSUM_TH = 1024.0f;  // a 4-key probability sum above this (some p > 2^8) triggers a rescale

#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
#define MFMA4(A, B, C) __builtin_amdgcn_mfma_f32_4x4x1f32((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// A 4-VGPR splat the compiler must keep materialised (it is the C operand of every score MFMA;
// rebuilding it with 4 v_mov per MFMA would cost as much VALU as the subtraction it replaces).
__device__ __forceinline__ f32x4 splat_opaque(float x) {
  f32x4 v = {x, x, x, x};
  asm volatile("" : "+v"(v));
  return v;
}

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

// Copy rows [r0, r1) (multiples of 16) of `nch` channel planes (global plane stride L) into LDS
// planes of stride Lp, computing mul * x, rows >= L filled with `fill`.
__device__ __forceinline__ void stage_planes(float* __restrict__ dst, int Lp, const float* __restrict__ src,
                                             int nch, int L, int r0, int r1, bool vec, float mul, float fill) {
  if (vec) {  // L % 4 == 0 and 16-byte aligned planes: a quad is entirely in or out
    const int nq = (r1 - r0) >> 2;
    for (int idx = threadIdx.x; idx < nch * nq; idx += blockDim.x) {
      const int c = idx / nq;
      const int m = r0 + 4 * (idx - c * nq);
      float4 t = make_float4(fill, fill, fill, fill);
      if (m < L) {
        t = *reinterpret_cast<const float4*>(src + (size_t)c * L + m);
        t.x *= mul; t.y *= mul; t.z *= mul; t.w *= mul;
      }
      *reinterpret_cast<float4*>(dst + c * Lp + m) = t;
    }
  } else {
    const int nr = r1 - r0;
    for (int idx = threadIdx.x; idx < nch * nr; idx += blockDim.x) {
      const int c = idx / nr;
      const int m = r0 + (idx - c * nr);
      dst[c * Lp + m] = m < L ? src[(size_t)c * L + m] * mul : fill;
    }
  }
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
__global__ void __launch_bounds__(512) attn_fwd_m44_kernel(const PgAttnArgs a) {
  extern __shared__ float4 lds4[];
  const int Lp = a.lp;
  float* kt = reinterpret_cast<float*>(lds4);  // K^T [4][Lp]
  float* vt = kt + 4 * Lp;                     // V^T [4][Lp]
  const Ids d = ids();
  const int lane = d.lane, qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int wg = gridDim.x - 1 - blockIdx.x;  // heaviest workgroups first
  const int first = wg * a.blocks_per_wg;
  const int nb = min(a.blocks_per_wg, NB - first);

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  float* op = a.o_out + (size_t)n * a.o_bs + (size_t)h * 4 * L;
  float* lsep = a.lse2_out + ((size_t)n * a.heads + h) * L;

  const int rows = 64 * (first + nb);
  stage_planes(kt, Lp, kp, 4, L, 0, rows, a.vec, 1.f, 0.f);
  stage_planes(vt, Lp, vp, 4, L, 0, rows, a.vec, 1.f, 0.f);
  __syncthreads();
  const int lo = first + d.wave, hi = first + nb - 1 - d.wave;
  if (lo > hi) return;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && lo == hi) break;
    const int blk = pass == 0 ? hi : lo;
    const int q0 = 64 * blk;
    const int ngrp = min(4, (L - q0 + 15) >> 4);  // 16-query groups of this block that exist

    float qf[4], mcur[4], lsum[4];
    f32x4 negm[4], acc[4];
    int qidx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qidx[t] = q0 + 16 * t + qi;
      qf[t] = qidx[t] < L ? qp[(size_t)g * L + qidx[t]] * a.scale2 : 0.f;
      mcur[t] = 0.f;  // provisional: the first tile (FORCE) moves every lane's m to its exact maximum
      negm[t] = splat_opaque(0.f);
      lsum[t] = 0.f;
      acc[t] = zero4;
    }

    // move group t's running max up (or, c < 0, down: first tile only) by c and rescale its state
    auto shift_max = [&](int t, float c) {
      const float alpha = ex2(-c);
      float al[4];
      quad_all(alpha, al);
      lsum[t] *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[t][i] *= al[i];
      mcur[t] += c;
      negm[t] = splat_opaque(-mcur[t]);
    };

    // One 16-key tile against groups [TMIN, 4): scores for all of them, exponentials, ONE overflow
    // test, then the P.V accumulation — 4 independent dependency chains per phase.
    //   MASK 0: no lane is cut (tile strictly below every diagonal)
    //   MASK 1: group TMIN is on its diagonal, per-lane causal predicate for it
    //   MASK 2: predicate for every group (the very first tile of block 0)
    //   FORCE : set every lane's m to the exact maximum of its allowed scores (first tile)
    auto step = [&](int k0, float kf, const f32x4& vf, auto TMIN_, auto MASK_, auto FORCE_) {
      constexpr int TMIN = decltype(TMIN_)::value, MASK = decltype(MASK_)::value;
      constexpr bool FORCE = decltype(FORCE_)::value;
      f32x4 s[4];
      float p[4][4], ps[4];
      int lim[4];
#pragma unroll
      for (int t = TMIN; t < 4; ++t) s[t] = MFMA16(kf, qf[t], negm[t]);
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
        // some score ran more than ~2^8 above its lane's running max: move the max up, rescale
#pragma unroll
        for (int t = TMIN; t < 4; ++t) {
          const bool cut = MASK == 2 || (MASK == 1 && t == TMIN);
          float c = NEG_BIG;
#pragma unroll
          for (int r = 0; r < 4; ++r) c = fmaxf(c, (!cut || r <= lim[t]) ? s[t][r] : NEG_BIG);
          if (cut) c = lim[t] >= 0 ? c : 0.f;  // no allowed key in this lane's subset: leave m alone
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
    auto frag_k = [&](int k0) { return kt[g * Lp + k0 + qi]; };
    auto frag_v = [&](int k0) { return *reinterpret_cast<const f32x4*>(vt + jc * Lp + k0 + 4 * g); };

    // ---- first tile: every lane that will ever see an allowed key sees one here (queries >= 16:
    //      the whole tile; queries < 16 have no other tile)
    if (q0 == 0) step(0, frag_k(0), frag_v(0), I<0>{}, I<2>{}, B<true>{});
    else step(0, frag_k(0), frag_v(0), I<0>{}, I<0>{}, B<true>{});

    // ---- full tiles (strictly below every group's diagonal): 4 groups per K/V fragment
    float kfn = frag_k(16);
    f32x4 vfn = frag_v(16);
    for (int k0 = 16; k0 < q0; k0 += 16) {
      const float kf = kfn;
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

    // ---- merge the four key subsets g of every query, normalise, store
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < ngrp) {
        const float m = lsum[t] > 0.f ? mcur[t] : NEG_BIG;  // a subset without allowed keys has no max
        float mx = fmaxf(m, __shfl_xor(m, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float f = ex2(m - mx);
        const float l = gsum(lsum[t] * f);
        float fq[4], iq[4], ov[4];
        quad_all(f, fq);
        quad_all(l > 0.f ? 1.f / l : 0.f, iq);
#pragma unroll
        for (int i = 0; i < 4; ++i) ov[i] = gsum(acc[t][i] * fq[i]) * iq[i];
        if (g == 0) {
          store_rows4(op + (size_t)jc * L, L, q0 + 16 * t + d.qb4, a.vec, ov[0], ov[1], ov[2], ov[3]);
          if (qidx[t] < L) lsep[qidx[t]] = l > 0.f ? mx + log2f(l) : POS_BIG;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ backward: dQ
// Also writes delta[l] = sum_j dO[l][j] * O[l][j] (read by the dK/dV kernel).
__global__ void __launch_bounds__(512) attn_dq_m44_kernel(const PgAttnArgs a) {
  extern __shared__ float4 lds4[];
  const int Lp = a.lp;
  float* kt = reinterpret_cast<float*>(lds4);
  float* vt = kt + 4 * Lp;
  const Ids d = ids();
  const int qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int wg = gridDim.x - 1 - blockIdx.x;
  const int first = wg * a.blocks_per_wg;
  const int nb = min(a.blocks_per_wg, NB - first);

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * 4 * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * 4 * L;
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * 4 * L;
  const size_t row = ((size_t)n * a.heads + h) * L;

  const int rows = 64 * (first + nb);
  stage_planes(kt, Lp, kp, 4, L, 0, rows, a.vec, 1.f, 0.f);
  stage_planes(vt, Lp, vp, 4, L, 0, rows, a.vec, 1.f, 0.f);
  __syncthreads();
  const int lo = first + d.wave, hi = first + nb - 1 - d.wave;
  if (lo > hi) return;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && lo == hi) break;
    const int blk = pass == 0 ? hi : lo;
    const int q0 = 64 * blk;
    const int ngrp = min(4, (L - q0 + 15) >> 4);

    float qf[4], gf[4];
    f32x4 nl[4], nd[4], acc[4];
    int qidx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      qidx[t] = q0 + 16 * t + qi;
      const bool ok = qidx[t] < L;
      const int qc = ok ? qidx[t] : L - 1;
      qf[t] = ok ? qp[(size_t)g * L + qc] * a.scale2 : 0.f;
      gf[t] = ok ? gp[(size_t)g * L + qc] : 0.f;
      const float ov = ok ? op[(size_t)g * L + qc] : 0.f;
      const float dl = gsum(gf[t] * ov);
      if (ok && g == 0) a.delta[row + qc] = dl;
      const float lse = ok ? a.lse2_in[row + qc] : POS_BIG;
      nl[t] = splat_opaque(-lse);
      nd[t] = splat_opaque(-dl);
      acc[t] = zero4;
    }

    // One 16-key tile against groups [TMIN, 4): P = exp2(S - lse), dS = P * (dP - delta), dQ += dS K.
    // MASK: group TMIN is on its diagonal (per-lane causal predicate).
    struct Frag { float kf, va; f32x4 kq; };
    auto frag = [&](int k0) {
      Frag f;
      f.kf = kt[g * Lp + k0 + qi];
      f.va = vt[g * Lp + k0 + qi];
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
        s[t] = MFMA16(f.kf, qf[t], nl[t]);
        dp[t] = MFMA16(f.va, gf[t], nd[t]);
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
__global__ void __launch_bounds__(512) attn_dkv_m44_kernel(const PgAttnArgs a) {
  extern __shared__ float4 lds4[];
  const int Lp = a.lp;
  float* qt = reinterpret_cast<float*>(lds4);  // Q^T  [4][Lp]
  float* gt = qt + 4 * Lp;                     // dO^T [4][Lp]
  float* nlse = gt + 4 * Lp;                   // -lse2 [Lp]   (-BIG for rows >= L: P = 0)
  float* ndel = nlse + Lp;                     // -delta [Lp]
  const Ids d = ids();
  const int qi = d.qi, g = d.g, jc = d.jc;
  const int h = blockIdx.y, n = blockIdx.z;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int first = blockIdx.x * a.blocks_per_wg;  // smallest keys (most queries) first
  const int nb = min(a.blocks_per_wg, NB - first);

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * 4 * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * 4 * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * 4 * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * 4 * L;
  float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * 4 * L;
  float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * 4 * L;
  const size_t row = ((size_t)n * a.heads + h) * L;

  const int r0 = 64 * first, r1 = 64 * NB;
  stage_planes(qt, Lp, qp, 4, L, r0, r1, a.vec, 1.f, 0.f);
  stage_planes(gt, Lp, gp, 4, L, r0, r1, a.vec, 1.f, 0.f);
  stage_planes(nlse, Lp, a.lse2_in + row, 1, L, r0, r1, a.vec, -1.f, NEG_BIG);
  stage_planes(ndel, Lp, a.delta + row, 1, L, r0, r1, a.vec, -1.f, 0.f);
  __syncthreads();
  const int lo = first + d.wave, hi = first + nb - 1 - d.wave;
  if (lo > hi) return;
  const int q_end = ((L + 15) >> 4) << 4;

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && lo == hi) break;
    const int blk = pass == 0 ? lo : hi;
    const int kb0 = 64 * blk;
    const int ngrp = min(4, (L - kb0 + 15) >> 4);

    float kf[4], vf[4];
    f32x4 acck[4], accv[4];
    int kidx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      kidx[t] = kb0 + 16 * t + qi;
      const bool ok = kidx[t] < L;
      const int kc = ok ? kidx[t] : L - 1;
      kf[t] = ok ? kp[(size_t)g * L + kc] * a.scale2 : 0.f;
      vf[t] = ok ? vp[(size_t)g * L + kc] : 0.f;
      acck[t] = zero4;
      accv[t] = zero4;
    }

    // One 16-query tile against key groups [0, TMAX]: P = exp2(S - lse), dS = P * (dP - delta),
    // dV += P^T dO, dK += dS^T Q. MASK: group TMAX is on its diagonal (per-lane causal predicate).
    struct Frag { float qa, ga; f32x4 cl, cd, qq, gq; };
    auto frag = [&](int q0t) {
      Frag f;
      f.qa = qt[g * Lp + q0t + qi];
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
        s[t] = MFMA16(f.qa, kf[t], f.cl);
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

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int pg_attn_mfma_launch(int which, const PgAttnArgs& a0, hipStream_t st) {
  if (a0.dk_dim != 4 || a0.dv_dim != 4) return 0;
  PgAttnArgs a = a0;
  const int NB = (a.L + 63) / 64;
  const int bpw = NB < 16 ? NB : 16;
  a.blocks_per_wg = bpw;
  a.lp = 64 * NB + 16;  // plane stride == 16 (mod 64): conflict-free b32 and b128 fragment reads
  const size_t planes = which == PG_ATTN_DKV ? 10 : 8;
  const size_t shmem = planes * (size_t)a.lp * sizeof(float);
  if (shmem > 160 * 1024) return 0;
  bool vec = a.L % 4 == 0;
  if (which == PG_ATTN_FWD)
    vec = vec && aligned16(a.k) && aligned16(a.v) && aligned16(a.o_out) && a.k_bs % 4 == 0 &&
          a.v_bs % 4 == 0 && a.o_bs % 4 == 0;
  else if (which == PG_ATTN_DQ)
    vec = vec && aligned16(a.k) && aligned16(a.v) && aligned16(a.dq) && a.k_bs % 4 == 0 &&
          a.v_bs % 4 == 0 && a.dq_bs % 4 == 0;
  else
    vec = vec && aligned16(a.q) && aligned16(a.d_o) && aligned16(a.lse2_in) && aligned16(a.delta) &&
          aligned16(a.dk) && aligned16(a.dv) && a.q_bs % 4 == 0 && a.do_bs % 4 == 0 &&
          a.dk_bs % 4 == 0 && a.dv_bs % 4 == 0;
  a.vec = vec ? 1 : 0;
  dim3 grid((unsigned)((NB + bpw - 1) / bpw), (unsigned)a.heads, (unsigned)a.N);
  dim3 block((unsigned)(64 * ((bpw + 1) / 2)));
  const void* fn = which == PG_ATTN_FWD  ? reinterpret_cast<const void*>(attn_fwd_m44_kernel)
                   : which == PG_ATTN_DQ ? reinterpret_cast<const void*>(attn_dq_m44_kernel)
                                         : reinterpret_cast<const void*>(attn_dkv_m44_kernel);
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  if (which == PG_ATTN_FWD)
    hipLaunchKernelGGL(attn_fwd_m44_kernel, grid, block, shmem, st, a);
  else if (which == PG_ATTN_DQ)
    hipLaunchKernelGGL(attn_dq_m44_kernel, grid, block, shmem, st, a);
  else
    hipLaunchKernelGGL(attn_dkv_m44_kernel, grid, block, shmem, st, a);
  return 1;
}
