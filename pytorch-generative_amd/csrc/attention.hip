// attention.hip — fused causal attention core for NCHW image tensors, fp32, gfx950.
//
// Reference: CausalAttention.forward nn/attention.py:147-160
//   attn = (q @ k^T) / sqrt(d_k); masked_fill(mask==0, -inf); softmax; masked_fill(mask==0, 0)
//   out  = attn @ v
// with mask = tril(ones(L, L), diagonal=-int(mask_center)) (nn/attention.py:60-63). The
// reference materialises five N*heads*L*L fp32 tensors per call and copies the L*L mask
// host->device on every forward; here nothing L*L ever exists (online softmax; the mask is
// the predicate key <= query - strict).
//
// CDNA4 mapping (head dims on this path are tiny: d_k = d_v = 4 for ImageGPT, 4/32 for
// PixelSNAIL — an MFMA tile would be mostly padding, and exp dominates):
//  * "row-owner" formulation: one lane owns QPL (=2) query rows (forward, dQ) or key rows
//    (dK/dV); q / o / running max+sum / gradients of its rows stay in VGPRs for the whole kernel.
//  * the opposite operand (keys+values in fwd/dQ, queries+dO+lse+delta in dK/dV) is streamed
//    through LDS in tiles of [KT rows][row-interleaved channels]; every lane of a wave reads the
//    SAME row at the same time, i.e. LDS broadcast reads (ds_read_b128, conflict free), and the
//    row is reused by both of the lane's own rows. (Measured on MI355X: the first version read the
//    streamed rows through the scalar cache (s_load_dwordx8 -> SGPR operands); it was bound by
//    scalar-load latency / issue — LDS tiles + 2 rows per lane run 1.4x faster.)
//  * loop bounds are wave-uniform (readfirstlane), so the causal triangle is skipped per wave
//    and only the diagonal band pays for per-lane predicates.
//  * exp via v_exp_f32 in the log2 domain (scale*log2(e) folded into q resp. k).
#include <stdlib.h>

#include "attention_args.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float POS_BIG = 1.0e30f;
constexpr int QPL = 2;  // rows owned per lane

using AttnArgs = PgAttnArgs;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DK, int DV> struct Cfg {
  static constexpr int RS = DK + DV;        // floats per staged key row (fwd / dQ)
  static constexpr int RS2 = DK + DV + 4;   // floats per staged query row (dK/dV): q|dO|lse|delta|pad
  static constexpr int CH = (RS <= 8) ? 8 : ((RS <= 16) ? 4 : 2);
  // dK/dV keeps 2x(dk + dv) accumulators per lane on top of the staged rows: a smaller chunk
  // keeps it under 96 VGPRs (5 waves/SIMD instead of 3 — measured: the kernel was occupancy bound)
  static constexpr int CH2 = (CH > 2) ? CH / 2 : 2;
};

// Stage KT rows [m0, m0+KT) of two channel-major sources (a: DA chans, b: DB chans) into the
// row-interleaved LDS tile. Global reads are coalesced along the row index; padded channels and
// rows >= L are zero filled.
template <int DA, int DB, int RS>
__device__ __forceinline__ void stage_rows(float* __restrict__ tile, const float* __restrict__ a,
                                           int da, const float* __restrict__ b, int db, int L,
                                           int m0, int KT, int tid, int nthreads) {
  for (int idx = tid; idx < (DA + DB) * KT; idx += nthreads) {
    const int c = idx / KT;
    const int m = idx - c * KT;
    const int gm = m0 + m;
    float val = 0.f;
    if (gm < L) {
      if (c < DA) {
        if (c < da) val = a[(size_t)c * L + gm];
      } else if (c - DA < db) {
        val = b[(size_t)(c - DA) * L + gm];
      }
    }
    tile[m * RS + c] = val;
  }
}

template <int D>
__device__ __forceinline__ void read_row(float (&dst)[D], const float* __restrict__ src) {
  static_assert(D % 4 == 0, "row pieces are float4 multiples");
#pragma unroll
  for (int i = 0; i < D / 4; ++i) {
    const float4 t = reinterpret_cast<const float4*>(src)[i];
    dst[4 * i] = t.x; dst[4 * i + 1] = t.y; dst[4 * i + 2] = t.z; dst[4 * i + 3] = t.w;
  }
}

// Row ownership and load balance: the 64-row blocks of a workgroup are handed out in PAIRS —
// wave w owns block bB = first + w and block bA = first + nb - 1 - w (slot A streams the whole
// range the wave needs, slot B only part of it) — so every wave of the workgroup streams the same
// number of rows of the causal triangle (measured: 1.35x over consecutive blocks). A lane holds
// one row of each block: slot 0 = A, slot 1 = B.
struct WavePlan {
  int bA, bB;    // 64-row block indices (bB < bA; bB == -1: the wave owns a single block)
  bool on;
};

__device__ __forceinline__ WavePlan plan_wave(int first, int nb, int wave) {
  WavePlan p;
  const int lo = first + wave, hi = first + nb - 1 - wave;
  p.on = lo <= hi;
  p.bA = hi;
  p.bB = lo < hi ? lo : -1;
  return p;
}

// ------------------------------------------------------------------------------ forward
// slot A = the later query block (more keys), slot B = the earlier one.
template <int DK, int DV, bool ACTB, bool MASKA, bool MASKB>
__device__ __forceinline__ void fwd_chunk(const float* __restrict__ rows, int mglob,
                                          const float (&qv)[QPL][DK], float (&acc)[QPL][DV],
                                          float (&mrun)[QPL], float (&lsum)[QPL],
                                          const int (&my_last)[QPL]) {
  constexpr int CH = Cfg<DK, DV>::CH, RS = Cfg<DK, DV>::RS;
  float kk[CH][DK], vv[CH][DV];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    read_row<DK>(kk[c], rows + c * RS);
    read_row<DV>(vv[c], rows + c * RS + DK);
  }
#pragma unroll
  for (int u = 0; u < (ACTB ? 2 : 1); ++u) {
    const bool masked = u == 0 ? MASKA : MASKB;
    float s[CH];
    float cmax = NEG_BIG;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = qv[u][0] * kk[c][0];
#pragma unroll
      for (int i = 1; i < DK; ++i) t = fmaf(qv[u][i], kk[c][i], t);
      if (masked) t = (mglob + c) <= my_last[u] ? t : NEG_BIG;
      s[c] = t;
      cmax = fmaxf(cmax, t);
    }
    const float mnew = fmaxf(mrun[u], cmax);
    const float alpha = fast_exp2(mrun[u] - mnew);
    lsum[u] *= alpha;
#pragma unroll
    for (int j = 0; j < DV; ++j) acc[u][j] *= alpha;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float p = fast_exp2(s[c] - mnew);
      if (masked) p = (mglob + c) <= my_last[u] ? p : 0.f;
      lsum[u] += p;
#pragma unroll
      for (int j = 0; j < DV; ++j) acc[u][j] = fmaf(p, vv[c][j], acc[u][j]);
    }
    mrun[u] = mnew;
  }
}

// key-range plan of a wave for the query-owning kernels (fwd, dQ); all bounds multiples of CH
// except endA.
struct KeyRegions { int fullB, endB, fullA, endA; };

template <int CH>
__device__ __forceinline__ KeyRegions key_regions(const WavePlan& w, int L, int strict) {
  KeyRegions r;
  if (!w.on) { r.fullB = r.endB = r.fullA = r.endA = 0; return r; }
  const int fa = 64 * w.bA - strict + 1;
  r.fullA = fa < 0 ? 0 : (fa / CH) * CH;                     // keys < fullA: allowed for all of A
  r.endA = min(64 * w.bA + 63, L - 1) - strict + 1;           // keys < endA: needed by some lane of A
  if (w.bB >= 0) {
    const int fb = 64 * w.bB - strict + 1;
    r.fullB = fb < 0 ? 0 : (fb / CH) * CH;
    r.endB = ((64 * w.bB + 63 - strict + 1 + CH - 1) / CH) * CH;  // rounded up (<= fullA)
  } else {
    r.fullB = r.endB = 0;
  }
  return r;
}

template <int DK, int DV>
__global__ void __launch_bounds__(512) attn_fwd_kernel(const AttnArgs a) {
  using C = Cfg<DK, DV>;
  extern __shared__ float4 lds4[];
  float* tile = reinterpret_cast<float*>(lds4);
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int wg = gridDim.x - 1 - blockIdx.x;  // heaviest workgroups first
  const int first = wg * a.blocks_per_wg;
  const int nb = min(a.blocks_per_wg, NB - first);
  const WavePlan w = plan_wave(first, nb, wave);

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * a.dk_dim * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * a.dk_dim * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * a.dv_dim * L;

  int lq[QPL], my_last[QPL];
  float qv[QPL][DK], acc[QPL][DV], mrun[QPL], lsum[QPL];
#pragma unroll
  for (int u = 0; u < QPL; ++u) {
    const int blk = u == 0 ? w.bA : w.bB;
    const bool slot_on = w.on && blk >= 0;
    lq[u] = slot_on ? 64 * blk + lane : L;    // L = "no row"
    const int lc = min(lq[u], L - 1);
    my_last[u] = lc - a.strict;
    mrun[u] = NEG_BIG;
    lsum[u] = 0.f;
#pragma unroll
    for (int i = 0; i < DK; ++i) qv[u][i] = (slot_on && i < a.dk_dim) ? qp[(size_t)i * L + lc] * a.scale2 : 0.f;
#pragma unroll
    for (int j = 0; j < DV; ++j) acc[u][j] = 0.f;
  }
  const KeyRegions r = key_regions<C::CH>(w, L, a.strict);
  const int blk_end = min(64 * (first + nb - 1) + 63, L - 1) - a.strict + 1;  // keys the workgroup needs

  for (int m0 = 0; m0 < blk_end; m0 += a.kt) {
    __syncthreads();
    stage_rows<DK, DV, C::RS>(tile, kp, a.dk_dim, vp, a.dv_dim, L, m0, a.kt, threadIdx.x, blockDim.x);
    __syncthreads();
    const int t_hi = m0 + a.kt;
    int m = m0;
    for (const int e = min(t_hi, r.fullB); m < e; m += C::CH)
      fwd_chunk<DK, DV, true, false, false>(tile + (m - m0) * C::RS, m, qv, acc, mrun, lsum, my_last);
    for (const int e = min(t_hi, r.endB); m < e; m += C::CH)
      fwd_chunk<DK, DV, true, false, true>(tile + (m - m0) * C::RS, m, qv, acc, mrun, lsum, my_last);
    for (const int e = min(t_hi, r.fullA); m < e; m += C::CH)
      fwd_chunk<DK, DV, false, false, false>(tile + (m - m0) * C::RS, m, qv, acc, mrun, lsum, my_last);
    for (const int e = min(t_hi, r.endA); m < e; m += C::CH)  // A's diagonal band
      fwd_chunk<DK, DV, false, true, false>(tile + (m - m0) * C::RS, m, qv, acc, mrun, lsum, my_last);
  }

#pragma unroll
  for (int u = 0; u < QPL; ++u) {
    if (lq[u] < L) {
      const float inv = lsum[u] > 0.f ? 1.f / lsum[u] : 0.f;
      float* op = a.o_out + (size_t)n * a.o_bs + (size_t)h * a.dv_dim * L + lq[u];
#pragma unroll
      for (int j = 0; j < DV; ++j)
        if (j < a.dv_dim) op[(size_t)j * L] = acc[u][j] * inv;
      a.lse2_out[((size_t)n * a.heads + h) * L + lq[u]] =
          lsum[u] > 0.f ? mrun[u] + log2f(lsum[u]) : POS_BIG;
    }
  }
}

// --------------------------------------------------------------------------- backward: dQ
template <int DK, int DV, bool ACTB, bool MASKA, bool MASKB>
__device__ __forceinline__ void dq_chunk(const float* __restrict__ rows, int mglob,
                                         const float (&qv)[QPL][DK], const float (&gv)[QPL][DV],
                                         float (&dqv)[QPL][DK], const float (&lse)[QPL],
                                         const float (&delta)[QPL], const int (&my_last)[QPL]) {
  constexpr int CH = Cfg<DK, DV>::CH, RS = Cfg<DK, DV>::RS;
  float kk[CH][DK], vv[CH][DV];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    read_row<DK>(kk[c], rows + c * RS);
    read_row<DV>(vv[c], rows + c * RS + DK);
  }
#pragma unroll
  for (int u = 0; u < (ACTB ? 2 : 1); ++u) {
    const bool masked = u == 0 ? MASKA : MASKB;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = qv[u][0] * kk[c][0];
#pragma unroll
      for (int i = 1; i < DK; ++i) t = fmaf(qv[u][i], kk[c][i], t);
      float dp = gv[u][0] * vv[c][0];
#pragma unroll
      for (int j = 1; j < DV; ++j) dp = fmaf(gv[u][j], vv[c][j], dp);
      float p = fast_exp2(t - lse[u]);
      if (masked) p = (mglob + c) <= my_last[u] ? p : 0.f;
      const float ds = p * (dp - delta[u]);
#pragma unroll
      for (int i = 0; i < DK; ++i) dqv[u][i] = fmaf(ds, kk[c][i], dqv[u][i]);
    }
  }
}

// lane = one query of block A and one of block B. Also writes delta[l] = sum_j dO[l,j]*O[l,j].
template <int DK, int DV>
__global__ void __launch_bounds__(512) attn_bwd_dq_kernel(const AttnArgs a) {
  using C = Cfg<DK, DV>;
  extern __shared__ float4 lds4[];
  float* tile = reinterpret_cast<float*>(lds4);
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int wg = gridDim.x - 1 - blockIdx.x;
  const int first = wg * a.blocks_per_wg;
  const int nb = min(a.blocks_per_wg, NB - first);
  const WavePlan w = plan_wave(first, nb, wave);

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * a.dk_dim * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * a.dk_dim * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * a.dv_dim * L;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * a.dv_dim * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * a.dv_dim * L;
  const size_t row = ((size_t)n * a.heads + h) * L;

  int lq[QPL], my_last[QPL];
  float qv[QPL][DK], dqv[QPL][DK], gv[QPL][DV], lse[QPL], delta[QPL];
#pragma unroll
  for (int u = 0; u < QPL; ++u) {
    const int blk = u == 0 ? w.bA : w.bB;
    const bool slot_on = w.on && blk >= 0;
    lq[u] = slot_on ? 64 * blk + lane : L;
    const int lc = min(lq[u], L - 1);
    my_last[u] = lc - a.strict;
#pragma unroll
    for (int i = 0; i < DK; ++i) {
      qv[u][i] = (slot_on && i < a.dk_dim) ? qp[(size_t)i * L + lc] * a.scale2 : 0.f;
      dqv[u][i] = 0.f;
    }
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      const bool ok = slot_on && j < a.dv_dim;
      gv[u][j] = ok ? gp[(size_t)j * L + lc] : 0.f;
      const float ov = ok ? op[(size_t)j * L + lc] : 0.f;
      d = fmaf(gv[u][j], ov, d);
    }
    delta[u] = d;
    lse[u] = slot_on ? a.lse2_in[row + lc] : POS_BIG;
    if (lq[u] < L) a.delta[row + lq[u]] = d;
  }
  const KeyRegions r = key_regions<C::CH>(w, L, a.strict);
  const int blk_end = min(64 * (first + nb - 1) + 63, L - 1) - a.strict + 1;

  for (int m0 = 0; m0 < blk_end; m0 += a.kt) {
    __syncthreads();
    stage_rows<DK, DV, C::RS>(tile, kp, a.dk_dim, vp, a.dv_dim, L, m0, a.kt, threadIdx.x, blockDim.x);
    __syncthreads();
    const int t_hi = m0 + a.kt;
    int m = m0;
    for (const int e = min(t_hi, r.fullB); m < e; m += C::CH)
      dq_chunk<DK, DV, true, false, false>(tile + (m - m0) * C::RS, m, qv, gv, dqv, lse, delta, my_last);
    for (const int e = min(t_hi, r.endB); m < e; m += C::CH)
      dq_chunk<DK, DV, true, false, true>(tile + (m - m0) * C::RS, m, qv, gv, dqv, lse, delta, my_last);
    for (const int e = min(t_hi, r.fullA); m < e; m += C::CH)
      dq_chunk<DK, DV, false, false, false>(tile + (m - m0) * C::RS, m, qv, gv, dqv, lse, delta, my_last);
    for (const int e = min(t_hi, r.endA); m < e; m += C::CH)
      dq_chunk<DK, DV, false, true, false>(tile + (m - m0) * C::RS, m, qv, gv, dqv, lse, delta, my_last);
  }
#pragma unroll
  for (int u = 0; u < QPL; ++u) {
    if (lq[u] < L) {
      float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * a.dk_dim * L + lq[u];
#pragma unroll
      for (int i = 0; i < DK; ++i)
        if (i < a.dk_dim) dqp[(size_t)i * L] = dqv[u][i] * a.scale;
    }
  }
}

// ------------------------------------------------------------------------ backward: dK, dV
// staged query row: [q(DK) | dO(DV) | lse2 | delta | pad pad]; rows >= L are neutral
// (q = dO = 0, lse = +BIG -> p = 0). slot A = the EARLIER key block (more queries), slot B = the
// later one.
template <int DK, int DV, bool ACTB, bool MASKA, bool MASKB>
__device__ __forceinline__ void dkv_chunk(const float* __restrict__ rows, int lglob,
                                          const float (&kv)[QPL][DK], const float (&vv)[QPL][DV],
                                          float (&dkv)[QPL][DK], float (&dvv)[QPL][DV],
                                          const int (&my_first)[QPL]) {
  constexpr int CH = Cfg<DK, DV>::CH2, RS2 = Cfg<DK, DV>::RS2;
  float qq[CH][DK], gg[CH][DV];
  float2 ld[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    read_row<DK>(qq[c], rows + c * RS2);
    read_row<DV>(gg[c], rows + c * RS2 + DK);
    ld[c] = *reinterpret_cast<const float2*>(rows + c * RS2 + DK + DV);
  }
#pragma unroll
  for (int u = 0; u < (ACTB ? 2 : 1); ++u) {
    const bool masked = u == 0 ? MASKA : MASKB;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = kv[u][0] * qq[c][0];
#pragma unroll
      for (int i = 1; i < DK; ++i) t = fmaf(kv[u][i], qq[c][i], t);
      float dp = vv[u][0] * gg[c][0];
#pragma unroll
      for (int j = 1; j < DV; ++j) dp = fmaf(vv[u][j], gg[c][j], dp);
      float p = fast_exp2(t - ld[c].x);
      if (masked) p = (lglob + c) >= my_first[u] ? p : 0.f;
      const float ds = p * (dp - ld[c].y);
#pragma unroll
      for (int j = 0; j < DV; ++j) dvv[u][j] = fmaf(p, gg[c][j], dvv[u][j]);
#pragma unroll
      for (int i = 0; i < DK; ++i) dkv[u][i] = fmaf(ds, qq[c][i], dkv[u][i]);
    }
  }
}

template <int DK, int DV>
__global__ void __launch_bounds__(512) attn_bwd_dkv_kernel(const AttnArgs a) {
  using C = Cfg<DK, DV>;
  constexpr int CH = C::CH2;
  extern __shared__ float4 lds4[];
  float* tile = reinterpret_cast<float*>(lds4);
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int L = a.L;
  const int NB = (L + 63) >> 6;
  const int first = blockIdx.x * a.blocks_per_wg;  // smallest keys (most queries) first
  const int nb = min(a.blocks_per_wg, NB - first);
  WavePlan w = plan_wave(first, nb, wave);
  // here the block that streams the whole range is the EARLIER one: swap roles
  if (w.on && w.bB >= 0) { const int t = w.bA; w.bA = w.bB; w.bB = t; }

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * a.dk_dim * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * a.dk_dim * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * a.dv_dim * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * a.dv_dim * L;
  const size_t row = ((size_t)n * a.heads + h) * L;
  const float* lsep = a.lse2_in + row;
  const float* dlp = a.delta + row;

  int mk[QPL], my_first[QPL];
  float kv[QPL][DK], dkv[QPL][DK], vv[QPL][DV], dvv[QPL][DV];
#pragma unroll
  for (int u = 0; u < QPL; ++u) {
    const int blk = u == 0 ? w.bA : w.bB;
    const bool slot_on = w.on && blk >= 0;
    mk[u] = slot_on ? 64 * blk + lane : L;
    const int mc = min(mk[u], L - 1);
    my_first[u] = mc + a.strict;
#pragma unroll
    for (int i = 0; i < DK; ++i) {
      kv[u][i] = (slot_on && i < a.dk_dim) ? kp[(size_t)i * L + mc] * a.scale2 : 0.f;
      dkv[u][i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < DV; ++j) {
      vv[u][j] = (slot_on && j < a.dv_dim) ? vp[(size_t)j * L + mc] : 0.f;
      dvv[u][j] = 0.f;
    }
  }
  // wave-uniform query regions (multiples of CH):
  //   [startA, fullA) A masked | [fullA, startB) A | [startB, fullB) A + B masked | [fullB, Lr) A + B
  const int Lr = ((L + CH - 1) / CH) * CH;
  int startA = Lr, fullA = Lr, startB = Lr, fullB = Lr;
  if (w.on) {
    startA = ((64 * w.bA + a.strict) / CH) * CH;
    fullA = min(Lr, ((min(64 * w.bA + 63, L - 1) + a.strict + CH - 1) / CH) * CH);
    if (w.bB >= 0) {
      startB = ((64 * w.bB + a.strict) / CH) * CH;
      fullB = min(Lr, ((min(64 * w.bB + 63, L - 1) + a.strict + CH - 1) / CH) * CH);
    }
  }
  const int blk_lo = ((64 * first + a.strict) / 64) * 64;  // first row the workgroup needs

  for (int t0 = blk_lo; t0 < L; t0 += a.kt2) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < (DK + DV) * a.kt2; idx += blockDim.x) {
      const int c = idx / a.kt2;
      const int m = idx - c * a.kt2;
      const int gl = t0 + m;
      float val = 0.f;
      if (gl < L) {
        if (c < DK) {
          if (c < a.dk_dim) val = qp[(size_t)c * L + gl];
        } else if (c - DK < a.dv_dim) {
          val = gp[(size_t)(c - DK) * L + gl];
        }
      }
      tile[m * C::RS2 + c] = val;
    }
    for (int m = threadIdx.x; m < a.kt2; m += blockDim.x) {
      const int gl = t0 + m;
      tile[m * C::RS2 + DK + DV] = gl < L ? lsep[gl] : POS_BIG;
      tile[m * C::RS2 + DK + DV + 1] = gl < L ? dlp[gl] : 0.f;
    }
    __syncthreads();
    const int t_hi = min(t0 + a.kt2, Lr);
    int lq = max(t0, startA);
    for (const int e = min(t_hi, fullA); lq < e; lq += CH)
      dkv_chunk<DK, DV, false, true, false>(tile + (lq - t0) * C::RS2, lq, kv, vv, dkv, dvv, my_first);
    for (const int e = min(t_hi, startB); lq < e; lq += CH)
      dkv_chunk<DK, DV, false, false, false>(tile + (lq - t0) * C::RS2, lq, kv, vv, dkv, dvv, my_first);
    for (const int e = min(t_hi, fullB); lq < e; lq += CH)
      dkv_chunk<DK, DV, true, false, true>(tile + (lq - t0) * C::RS2, lq, kv, vv, dkv, dvv, my_first);
    for (; lq < t_hi; lq += CH)
      dkv_chunk<DK, DV, true, false, false>(tile + (lq - t0) * C::RS2, lq, kv, vv, dkv, dvv, my_first);
  }

#pragma unroll
  for (int u = 0; u < QPL; ++u) {
    if (mk[u] < L) {
      float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * a.dk_dim * L + mk[u];
      float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * a.dv_dim * L + mk[u];
#pragma unroll
      for (int i = 0; i < DK; ++i)
        if (i < a.dk_dim) dkp[(size_t)i * L] = dkv[u][i] * a.scale;
#pragma unroll
      for (int j = 0; j < DV; ++j)
        if (j < a.dv_dim) dvp[(size_t)j * L] = dvv[u][j];
    }
  }
}

enum { K_FWD = PG_ATTN_FWD, K_DQ = PG_ATTN_DQ, K_DKV = PG_ATTN_DKV };

template <int DK, int DV>
void launch_one(int which, const AttnArgs& a, dim3 grid, dim3 block, hipStream_t st) {
  using C = Cfg<DK, DV>;
  // tile rows: everything the workgroup streams if it fits ~64 KB of LDS, else 64-row multiples
  AttnArgs b = a;
  const int rows = ((b.L + 63) / 64) * 64;
  const int budget = 16384;  // floats
  int kt = (budget / C::RS / 64) * 64, kt2 = (budget / C::RS2 / 64) * 64;
  b.kt = kt < 64 ? 64 : (kt > rows ? rows : kt);
  b.kt2 = kt2 < 64 ? 64 : (kt2 > rows ? rows : kt2);
  if (which == K_FWD)
    hipLaunchKernelGGL((attn_fwd_kernel<DK, DV>), grid, block, (size_t)b.kt * C::RS * sizeof(float), st, b);
  else if (which == K_DQ)
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DK, DV>), grid, block, (size_t)b.kt * C::RS * sizeof(float), st, b);
  else
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DK, DV>), grid, block, (size_t)b.kt2 * C::RS2 * sizeof(float), st, b);
}

// Instantiated head dims of these VALU kernels (padded up): d_k <= 4 with d_v <= 32, d_k <= 16 with d_v <= 16 — the
// register-resident "row owner" formulation holds 2 x (d_k + d_v) operand values (+ as many accumulators in dK/dV) per
// lane, which beyond these sizes no longer fits the register file (round 3 shipped <64, 64> with 1 755 spilled
// registers). Larger head dims run on the matrix-core kernels of attention_k4.hip (multiples of 16, L % 16 == 0);
// ops.causal_attention zero-pads other shapes to the next such size.
template <int DK>
int launch_dv(int which, const AttnArgs& a, dim3 grid, dim3 block, hipStream_t st) {
  const int dv = a.dv_dim;
  if (dv <= 4) launch_one<DK, 4>(which, a, grid, block, st);
  else if (dv <= 16) launch_one<DK, 16>(which, a, grid, block, st);
  else if (DK == 4 && dv <= 32) launch_one<4, 32>(which, a, grid, block, st);
  else return PG_ESHAPE;
  return 0;
}

int launch_attn(int which, AttnArgs& a, hipStream_t st) {
  // a workgroup owns up to 16 consecutive 64-row blocks, handed to its waves in balanced pairs;
  // small L -> one workgroup per (n, head)
  const int NB = (a.L + 63) / 64;
  const int bpw = NB < 16 ? NB : 16;
  a.blocks_per_wg = bpw;
  const int nwaves = (bpw + 1) / 2;
  dim3 grid((unsigned)((NB + bpw - 1) / bpw), (unsigned)a.heads, (unsigned)a.N);
  dim3 block((unsigned)(64 * nwaves));
  const int dk = a.dk_dim;
  if (dk <= 4) return launch_dv<4>(which, a, grid, block, st);
  if (dk <= 16) return launch_dv<16>(which, a, grid, block, st);
  return PG_ESHAPE;
}

int check_dims(const char* who, int N, int heads, int L, int dk, int dv, int strict) {
  PG_REQUIRE(N > 0 && heads > 0 && L > 0 && dk > 0 && dv > 0, PG_EINVAL, "%s: non-positive dim", who);
  PG_REQUIRE(N <= 65535 && heads <= 65535, PG_ESHAPE, "%s: N/heads exceed grid limits", who);
  PG_REQUIRE(dk <= 64 && dv <= 64, PG_ESHAPE, "%s: head dims (%d,%d) > 64 unsupported", who, dk, dv);
  PG_REQUIRE(strict == 0 || strict == 1, PG_EINVAL, "%s: strict must be 0/1", who);
  return 0;
}

}  // namespace

// d_k = d_v = 4 (ImageGPT) runs on the matrix cores (attention_mfma.hip); PG_ATTN_MFMA=0 forces the
// VALU row-owner kernels above (A/B measurements, tests of both paths).
static bool use_mfma_attention() {
  static const bool on = []() {
    const char* e = PG_AB_ENV("PG_ATTN_MFMA");
    return !(e && e[0] == '0');
  }();
  return on;
}

// launch `which` on whichever kernel family covers the shape
static int launch_any(int which, AttnArgs& a, hipStream_t st) {
  // ab library: PG_ATTN_MFMA_MASK = bit mask of the kernels allowed on the matrix-core family (1 forward, 2 dQ, 4 dK/dV)
  static const int mask = []() { const char* e = PG_AB_ENV("PG_ATTN_MFMA_MASK"); return e ? atoi(e) : 7; }();
  const int bit = which == K_FWD ? 1 : (which == K_DQ ? 2 : 4);
  if (use_mfma_attention() && (mask & bit) && pg_attn_mfma_launch(which, a, st) == 1) return 0;
  return launch_attn(which, a, st);
}

PG_EXPORT int pg_causal_attn_fwd(const float* q, const float* k, const float* v, float* o,
                                 float* lse2, int N, int heads, int L, int dk, int dv, long q_bs,
                                 long k_bs, long v_bs, long o_bs, int strict, void* stream) {
  PG_REQUIRE(q && k && v && o && lse2, PG_EINVAL, "pg_causal_attn_fwd: null pointer");
  int rc = check_dims("pg_causal_attn_fwd", N, heads, L, dk, dv, strict);
  if (rc) return rc;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o_out = o; a.lse2_out = lse2;
  a.N = N; a.heads = heads; a.L = L; a.dk_dim = dk; a.dv_dim = dv; a.strict = strict;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs;
  a.scale = 1.f / sqrtf((float)dk);
  a.scale2 = a.scale * 1.44269504088896340736f;
  rc = launch_any(K_FWD, a, (hipStream_t)stream);
  PG_REQUIRE(rc == 0, rc, "pg_causal_attn_fwd: head dims (%d, %d) at L = %d are not instantiated: pad them to multiples of 16 and L to a multiple of 16 (ops.causal_attention does)", dk, dv, L);
  PG_LAUNCH_CHECK("pg_causal_attn_fwd");
  return 0;
}

namespace {
int attn_bwd_impl(int which_mask, const float* q, const float* k, const float* v, const float* o,
                  const float* d_o, const float* lse2, float* delta, float* dq, float* dk, float* dv,
                  int N, int heads, int L, int dk_dim, int dv_dim, long q_bs, long k_bs, long v_bs,
                  long o_bs, long do_bs, long dq_bs, long dk_bs, long dv_bs, int strict, void* stream) {
  PG_REQUIRE(q && k && v && o && d_o && lse2 && delta && dq && dk && dv, PG_EINVAL,
             "pg_causal_attn_bwd: null pointer");
  int rc = check_dims("pg_causal_attn_bwd", N, heads, L, dk_dim, dv_dim, strict);
  if (rc) return rc;
  AttnArgs a = {};
  a.q = q; a.k = k; a.v = v; a.o = o; a.d_o = d_o; a.lse2_in = lse2; a.delta = delta;
  a.dq = dq; a.dk = dk; a.dv = dv;
  a.N = N; a.heads = heads; a.L = L; a.dk_dim = dk_dim; a.dv_dim = dv_dim; a.strict = strict;
  a.q_bs = q_bs; a.k_bs = k_bs; a.v_bs = v_bs; a.o_bs = o_bs; a.do_bs = do_bs;
  a.dq_bs = dq_bs; a.dk_bs = dk_bs; a.dv_bs = dv_bs;
  a.scale = 1.f / sqrtf((float)dk_dim);
  a.scale2 = a.scale * 1.44269504088896340736f;
  if (which_mask == 3 && use_mfma_attention() && pg_attn_mfma_launch(PG_ATTN_BWD, a, (hipStream_t)stream) == 1) {
    PG_LAUNCH_CHECK("pg_causal_attn_bwd(fused)");  // dQ, dK, dV in one pass (attention_mfma.hip)
    return 0;
  }
  if (which_mask & 1) {
    rc = launch_any(K_DQ, a, (hipStream_t)stream);
    PG_REQUIRE(rc == 0, rc, "pg_causal_attn_bwd: head dims (%d, %d) at L = %d are not instantiated (see pg_causal_attn_fwd)", dk_dim, dv_dim, L);
    PG_LAUNCH_CHECK("pg_causal_attn_bwd(dq)");
  }
  if (which_mask & 2) {
    rc = launch_any(K_DKV, a, (hipStream_t)stream);
    PG_REQUIRE(rc == 0, rc, "pg_causal_attn_bwd: head dims (%d, %d) at L = %d are not instantiated (see pg_causal_attn_fwd)", dk_dim, dv_dim, L);
    PG_LAUNCH_CHECK("pg_causal_attn_bwd(dkv)");
  }
  return 0;
}
}  // namespace

#define PG_ATTN_BWD_ARGS                                                                          \
  q, k, v, o, d_o, lse2, delta, dq, dk, dv, N, heads, L, dk_dim, dv_dim, q_bs, k_bs, v_bs, o_bs,  \
      do_bs, dq_bs, dk_bs, dv_bs, strict, stream

PG_EXPORT int pg_causal_attn_bwd(const float* q, const float* k, const float* v, const float* o,
                                 const float* d_o, const float* lse2, float* delta, float* dq,
                                 float* dk, float* dv, int N, int heads, int L, int dk_dim,
                                 int dv_dim, long q_bs, long k_bs, long v_bs, long o_bs, long do_bs,
                                 long dq_bs, long dk_bs, long dv_bs, int strict, void* stream) {
  return attn_bwd_impl(3, PG_ATTN_BWD_ARGS);
}

// The two launches of pg_causal_attn_bwd individually (profiling / roofline measurement):
// _dq writes dq and delta; _dkv reads delta and writes dk, dv.
PG_EXPORT int pg_causal_attn_bwd_dq(const float* q, const float* k, const float* v, const float* o,
                                    const float* d_o, const float* lse2, float* delta, float* dq,
                                    float* dk, float* dv, int N, int heads, int L, int dk_dim,
                                    int dv_dim, long q_bs, long k_bs, long v_bs, long o_bs,
                                    long do_bs, long dq_bs, long dk_bs, long dv_bs, int strict,
                                    void* stream) {
  return attn_bwd_impl(1, PG_ATTN_BWD_ARGS);
}

PG_EXPORT int pg_causal_attn_bwd_dkv(const float* q, const float* k, const float* v, const float* o,
                                     const float* d_o, const float* lse2, float* delta, float* dq,
                                     float* dk, float* dv, int N, int heads, int L, int dk_dim,
                                     int dv_dim, long q_bs, long k_bs, long v_bs, long o_bs,
                                     long do_bs, long dq_bs, long dk_bs, long dv_bs, int strict,
                                     void* stream) {
  return attn_bwd_impl(2, PG_ATTN_BWD_ARGS);
}
