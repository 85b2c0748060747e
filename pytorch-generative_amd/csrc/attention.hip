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
// CDNA4 mapping (head dims here are tiny: d_k = d_v = 4 for ImageGPT, 4/32 for PixelSNAIL, so
// an MFMA tile would be mostly padding):
//  * one lane owns one query row (forward, dQ) or one key row (dK/dV); its q / o / running
//    max+sum live in VGPRs for the whole kernel.
//  * the row being streamed (keys in fwd/dQ, queries in dK/dV) is WAVE-UNIFORM, so it is
//    read through the scalar unit (s_load_dwordxN straight from the channel-major (d, L)
//    layout the 1x1 convs emit) and enters the FMAs as SGPR operands: no LDS, no barriers,
//    no transposes. Loop bounds are wave-uniform (readfirstlane) so the causal triangle is
//    skipped per wave, and only the 64-wide diagonal band pays for per-lane predicates.
//  * exp via v_exp_f32 in the log2 domain (scale*log2(e) folded into q).
#include "common.h"

namespace {

constexpr int AT_THREADS = 256;
constexpr float NEG_BIG = -1.0e30f;
constexpr float POS_BIG = 1.0e30f;

struct AttnArgs {
  const float* q; const float* k; const float* v; const float* o; const float* d_o;
  const float* lse2_in;
  float* o_out; float* lse2_out; float* delta; float* dq; float* dk; float* dv;
  int N, heads, L, dk_dim, dv_dim, strict;
  long q_bs, k_bs, v_bs, o_bs, do_bs, dq_bs, dk_bs, dv_bs;
  float scale, scale2;  // 1/sqrt(dk), log2(e)/sqrt(dk)
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DK, int DV> struct ChunkOf {
  static constexpr int value = (DK + DV <= 8) ? 8 : ((DK + DV <= 16) ? 4 : 2);
};

// Load CH consecutive positions of D channel rows (row stride L) starting at `pos`. The
// addresses are wave-uniform, so this becomes D s_load_dwordx{CH} instructions. Padded
// channels (i >= dim) re-read the last real row (their multiplier is zero / result unused).
template <int D, int CH, bool CLAMP>
__device__ __forceinline__ void load_rows(float (&dst)[D][CH], const float* __restrict__ base,
                                          int dim, int L, int pos) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const float* r = base + (size_t)(i < dim ? i : dim - 1) * L;
    if (CLAMP) {
#pragma unroll
      for (int c = 0; c < CH; ++c) dst[i][c] = r[min(pos + c, L - 1)];
    } else {
      const float* rp = r + pos;
#pragma unroll
      for (int c = 0; c < CH; ++c) dst[i][c] = rp[c];
    }
  }
}

template <int CH, bool CLAMP>
__device__ __forceinline__ void load_row1(float (&dst)[CH], const float* __restrict__ r, int L,
                                          int pos) {
#pragma unroll
  for (int c = 0; c < CH; ++c) dst[c] = CLAMP ? r[min(pos + c, L - 1)] : r[pos + c];
}

// ------------------------------------------------------------------------------ forward
template <int DK, int DV, int CH, bool MASKED, bool CLAMP>
__device__ __forceinline__ void fwd_chunk(const float (&qv)[DK], float (&acc)[DV], float& mrun,
                                          float& lsum, const float* __restrict__ kp,
                                          const float* __restrict__ vp, int dk, int dv, int L,
                                          int m, int my_last) {
  float kk[DK][CH], vv[DV][CH];
  load_rows<DK, CH, CLAMP>(kk, kp, dk, L, m);
  load_rows<DV, CH, CLAMP>(vv, vp, dv, L, m);
  float s[CH];
  float cmax = NEG_BIG;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < DK; ++i) t = fmaf(qv[i], kk[i][c], t);
    if (MASKED) t = (m + c) <= my_last ? t : NEG_BIG;
    s[c] = t;
    cmax = fmaxf(cmax, t);
  }
  const float mnew = fmaxf(mrun, cmax);
  const float alpha = fast_exp2(mrun - mnew);
  lsum *= alpha;
#pragma unroll
  for (int j = 0; j < DV; ++j) acc[j] *= alpha;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    float p = fast_exp2(s[c] - mnew);
    if (MASKED) p = (m + c) <= my_last ? p : 0.f;
    lsum += p;
#pragma unroll
    for (int j = 0; j < DV; ++j) acc[j] = fmaf(p, vv[j][c], acc[j]);
  }
  mrun = mnew;
}

template <int DK, int DV>
__global__ void __launch_bounds__(AT_THREADS) attn_fwd_kernel(const AttnArgs a) {
  constexpr int CH = ChunkOf<DK, DV>::value;
  const int qb = gridDim.x - 1 - blockIdx.x;  // heaviest (largest l) blocks first
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L;
  const int l0 = qb * AT_THREADS + wave * 64;
  if (l0 >= L) return;
  const int l = l0 + (threadIdx.x & 63);
  const bool valid = l < L;
  const int lc = valid ? l : L - 1;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * a.dk_dim * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * a.dk_dim * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * a.dv_dim * L;

  float qv[DK];
#pragma unroll
  for (int i = 0; i < DK; ++i) qv[i] = i < a.dk_dim ? qp[(size_t)i * L + lc] * a.scale2 : 0.f;

  float mrun = NEG_BIG, lsum = 0.f;
  float acc[DV];
#pragma unroll
  for (int j = 0; j < DV; ++j) acc[j] = 0.f;

  const int lmax = min(l0 + 63, L - 1);
  const int m_end = lmax - a.strict + 1;            // keys [0, m_end) are needed by some lane
  int m_full = l0 - a.strict + 1;                   // keys [0, m_full) are allowed for every lane
  m_full = m_full < 0 ? 0 : (m_full / CH) * CH;
  const int my_last = lc - a.strict;                // last allowed key of this lane

  int m = 0;
  for (; m < m_full; m += CH)
    fwd_chunk<DK, DV, CH, false, false>(qv, acc, mrun, lsum, kp, vp, a.dk_dim, a.dv_dim, L, m, my_last);
  // diagonal band: per-lane predicate
  for (; m + CH <= L && m < m_end; m += CH)
    fwd_chunk<DK, DV, CH, true, false>(qv, acc, mrun, lsum, kp, vp, a.dk_dim, a.dv_dim, L, m, my_last);
  // ragged last chunk (L % CH != 0): clamped (still wave-uniform) indices
  for (; m < m_end; m += CH)
    fwd_chunk<DK, DV, CH, true, true>(qv, acc, mrun, lsum, kp, vp, a.dk_dim, a.dv_dim, L, m, my_last);

  if (!valid) return;
  const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
  float* op = a.o_out + (size_t)n * a.o_bs + (size_t)h * a.dv_dim * L + l;
#pragma unroll
  for (int j = 0; j < DV; ++j)
    if (j < a.dv_dim) op[(size_t)j * L] = acc[j] * inv;
  a.lse2_out[((size_t)n * a.heads + h) * L + l] = lsum > 0.f ? mrun + log2f(lsum) : POS_BIG;
}

// --------------------------------------------------------------------------- backward: dQ
template <int DK, int DV, int CH, bool MASKED, bool CLAMP>
__device__ __forceinline__ void dq_chunk(const float (&qv)[DK], const float (&gv)[DV],
                                         float (&dqv)[DK], float lse, float delta,
                                         const float* __restrict__ kp,
                                         const float* __restrict__ vp, int dk, int dv, int L,
                                         int m, int my_last) {
  float kk[DK][CH], vv[DV][CH];
  load_rows<DK, CH, CLAMP>(kk, kp, dk, L, m);
  load_rows<DV, CH, CLAMP>(vv, vp, dv, L, m);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    float t = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < DK; ++i) t = fmaf(qv[i], kk[i][c], t);
#pragma unroll
    for (int j = 0; j < DV; ++j) dp = fmaf(gv[j], vv[j][c], dp);
    float p = fast_exp2(t - lse);
    if (MASKED) p = (m + c) <= my_last ? p : 0.f;
    const float ds = p * (dp - delta);
#pragma unroll
    for (int i = 0; i < DK; ++i) dqv[i] = fmaf(ds, kk[i][c], dqv[i]);
  }
}

// lane = query. Also writes delta[l] = sum_j do[l,j]*o[l,j] for the dK/dV pass.
template <int DK, int DV>
__global__ void __launch_bounds__(AT_THREADS) attn_bwd_dq_kernel(const AttnArgs a) {
  constexpr int CH = ChunkOf<DK, DV>::value;
  const int qb = gridDim.x - 1 - blockIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L;
  const int l0 = qb * AT_THREADS + wave * 64;
  if (l0 >= L) return;
  const int l = l0 + (threadIdx.x & 63);
  const bool valid = l < L;
  const int lc = valid ? l : L - 1;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * a.dk_dim * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * a.dk_dim * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * a.dv_dim * L;
  const float* op = a.o + (size_t)n * a.o_bs + (size_t)h * a.dv_dim * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * a.dv_dim * L;

  float qv[DK], dqv[DK], gv[DV];
#pragma unroll
  for (int i = 0; i < DK; ++i) {
    qv[i] = i < a.dk_dim ? qp[(size_t)i * L + lc] * a.scale2 : 0.f;
    dqv[i] = 0.f;
  }
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < DV; ++j) {
    gv[j] = j < a.dv_dim ? gp[(size_t)j * L + lc] : 0.f;
    const float ov = j < a.dv_dim ? op[(size_t)j * L + lc] : 0.f;
    delta = fmaf(gv[j], ov, delta);
  }
  const size_t row = ((size_t)n * a.heads + h) * L;
  const float lse = a.lse2_in[row + lc];
  if (valid) a.delta[row + l] = delta;

  const int lmax = min(l0 + 63, L - 1);
  const int m_end = lmax - a.strict + 1;
  int m_full = l0 - a.strict + 1;
  m_full = m_full < 0 ? 0 : (m_full / CH) * CH;
  const int my_last = lc - a.strict;

  int m = 0;
  for (; m < m_full; m += CH)
    dq_chunk<DK, DV, CH, false, false>(qv, gv, dqv, lse, delta, kp, vp, a.dk_dim, a.dv_dim, L, m, my_last);
  for (; m + CH <= L && m < m_end; m += CH)
    dq_chunk<DK, DV, CH, true, false>(qv, gv, dqv, lse, delta, kp, vp, a.dk_dim, a.dv_dim, L, m, my_last);
  for (; m < m_end; m += CH)
    dq_chunk<DK, DV, CH, true, true>(qv, gv, dqv, lse, delta, kp, vp, a.dk_dim, a.dv_dim, L, m, my_last);

  if (!valid) return;
  float* dqp = a.dq + (size_t)n * a.dq_bs + (size_t)h * a.dk_dim * L + l;
#pragma unroll
  for (int i = 0; i < DK; ++i)
    if (i < a.dk_dim) dqp[(size_t)i * L] = dqv[i] * a.scale;
}

// ------------------------------------------------------------------------ backward: dK, dV
template <int DK, int DV, int CH, bool MASKED, bool CLAMP>
__device__ __forceinline__ void dkv_chunk(const float (&kv)[DK], const float (&vv)[DV],
                                          float (&dkv)[DK], float (&dvv)[DV],
                                          const float* __restrict__ qp,
                                          const float* __restrict__ gp,
                                          const float* __restrict__ lsep,
                                          const float* __restrict__ dlp, int dk, int dv, int L,
                                          int lq, int my_first) {
  float qq[DK][CH], gg[DV][CH], ls[CH], dl[CH];
  load_rows<DK, CH, CLAMP>(qq, qp, dk, L, lq);
  load_rows<DV, CH, CLAMP>(gg, gp, dv, L, lq);
  load_row1<CH, CLAMP>(ls, lsep, L, lq);
  load_row1<CH, CLAMP>(dl, dlp, L, lq);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    float t = 0.f, dp = 0.f;
#pragma unroll
    for (int i = 0; i < DK; ++i) t = fmaf(kv[i], qq[i][c], t);
#pragma unroll
    for (int j = 0; j < DV; ++j) dp = fmaf(vv[j], gg[j][c], dp);
    float p = fast_exp2(t - ls[c]);
    if (MASKED) p = ((lq + c) >= my_first && (lq + c) < L) ? p : 0.f;
    const float ds = p * (dp - dl[c]);
#pragma unroll
    for (int j = 0; j < DV; ++j) dvv[j] = fmaf(p, gg[j][c], dvv[j]);
#pragma unroll
    for (int i = 0; i < DK; ++i) dkv[i] = fmaf(ds, qq[i][c], dkv[i]);
  }
}

// lane = key m; streams the queries l >= m + strict through the scalar unit.
template <int DK, int DV>
__global__ void __launch_bounds__(AT_THREADS) attn_bwd_dkv_kernel(const AttnArgs a) {
  constexpr int CH = ChunkOf<DK, DV>::value;
  const int kb = blockIdx.x;  // smallest keys (most queries) first
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L;
  const int m0 = kb * AT_THREADS + wave * 64;
  if (m0 >= L) return;
  const int mkey = m0 + (threadIdx.x & 63);
  const bool valid = mkey < L;
  const int mc_ = valid ? mkey : L - 1;

  const float* qp = a.q + (size_t)n * a.q_bs + (size_t)h * a.dk_dim * L;
  const float* kp = a.k + (size_t)n * a.k_bs + (size_t)h * a.dk_dim * L;
  const float* vp = a.v + (size_t)n * a.v_bs + (size_t)h * a.dv_dim * L;
  const float* gp = a.d_o + (size_t)n * a.do_bs + (size_t)h * a.dv_dim * L;
  const size_t row = ((size_t)n * a.heads + h) * L;
  const float* lsep = a.lse2_in + row;
  const float* dlp = a.delta + row;

  float kv[DK], dkv[DK], vv[DV], dvv[DV];
#pragma unroll
  for (int i = 0; i < DK; ++i) {
    kv[i] = i < a.dk_dim ? kp[(size_t)i * L + mc_] * a.scale2 : 0.f;
    dkv[i] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < DV; ++j) {
    vv[j] = j < a.dv_dim ? vp[(size_t)j * L + mc_] : 0.f;
    dvv[j] = 0.f;
  }

  // queries l >= key + strict. Band [m0+strict, l_full) needs predicates; [l_full, L) is
  // allowed for every lane of the wave.
  const int my_first = mc_ + a.strict;
  int lq = ((m0 + a.strict) / CH) * CH;
  int l_full = ((m0 + 63 + a.strict + CH - 1) / CH) * CH;
  const int l_band_end = l_full < L ? l_full : L;
  const int L_full_end = (L / CH) * CH;

  for (; lq + CH <= L && lq < l_band_end; lq += CH)
    dkv_chunk<DK, DV, CH, true, false>(kv, vv, dkv, dvv, qp, gp, lsep, dlp, a.dk_dim, a.dv_dim, L, lq, my_first);
  for (; lq < L_full_end; lq += CH)
    dkv_chunk<DK, DV, CH, false, false>(kv, vv, dkv, dvv, qp, gp, lsep, dlp, a.dk_dim, a.dv_dim, L, lq, my_first);
  // ragged tail (L % CH != 0)
  for (; lq < L; lq += CH)
    dkv_chunk<DK, DV, CH, true, true>(kv, vv, dkv, dvv, qp, gp, lsep, dlp, a.dk_dim, a.dv_dim, L, lq, my_first);

  if (!valid) return;
  float* dkp = a.dk + (size_t)n * a.dk_bs + (size_t)h * a.dk_dim * L + mkey;
  float* dvp = a.dv + (size_t)n * a.dv_bs + (size_t)h * a.dv_dim * L + mkey;
#pragma unroll
  for (int i = 0; i < DK; ++i)
    if (i < a.dk_dim) dkp[(size_t)i * L] = dkv[i] * a.scale;
#pragma unroll
  for (int j = 0; j < DV; ++j)
    if (j < a.dv_dim) dvp[(size_t)j * L] = dvv[j];
}

enum { K_FWD = 0, K_DQ = 1, K_DKV = 2 };

template <int DK, int DV>
void launch_one(int which, const AttnArgs& a, dim3 grid, hipStream_t st) {
  if (which == K_FWD) hipLaunchKernelGGL((attn_fwd_kernel<DK, DV>), grid, dim3(AT_THREADS), 0, st, a);
  else if (which == K_DQ) hipLaunchKernelGGL((attn_bwd_dq_kernel<DK, DV>), grid, dim3(AT_THREADS), 0, st, a);
  else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DK, DV>), grid, dim3(AT_THREADS), 0, st, a);
}

template <int DK>
int launch_dv(int which, const AttnArgs& a, dim3 grid, hipStream_t st) {
  const int dv = a.dv_dim;
  if (dv <= 4) launch_one<DK, 4>(which, a, grid, st);
  else if (dv <= 16) launch_one<DK, 16>(which, a, grid, st);
  else if (dv <= 32) launch_one<DK, 32>(which, a, grid, st);
  else if (dv <= 64) launch_one<DK, 64>(which, a, grid, st);
  else return PG_ESHAPE;
  return 0;
}

int launch_attn(int which, const AttnArgs& a, hipStream_t st) {
  dim3 grid((unsigned)((a.L + AT_THREADS - 1) / AT_THREADS), (unsigned)a.heads, (unsigned)a.N);
  const int dk = a.dk_dim;
  if (dk <= 4) return launch_dv<4>(which, a, grid, st);
  if (dk <= 16) return launch_dv<16>(which, a, grid, st);
  if (dk <= 64) return launch_dv<64>(which, a, grid, st);
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
  rc = launch_attn(K_FWD, a, (hipStream_t)stream);
  PG_REQUIRE(rc == 0, rc, "pg_causal_attn_fwd: unsupported head dims");
  PG_LAUNCH_CHECK("pg_causal_attn_fwd");
  return 0;
}

PG_EXPORT int pg_causal_attn_bwd(const float* q, const float* k, const float* v, const float* o,
                                 const float* d_o, const float* lse2, float* delta, float* dq,
                                 float* dk, float* dv, int N, int heads, int L, int dk_dim,
                                 int dv_dim, long q_bs, long k_bs, long v_bs, long o_bs, long do_bs,
                                 long dq_bs, long dk_bs, long dv_bs, int strict, void* stream) {
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
  rc = launch_attn(K_DQ, a, (hipStream_t)stream);
  PG_REQUIRE(rc == 0, rc, "pg_causal_attn_bwd: unsupported head dims");
  PG_LAUNCH_CHECK("pg_causal_attn_bwd(dq)");
  rc = launch_attn(K_DKV, a, (hipStream_t)stream);
  PG_REQUIRE(rc == 0, rc, "pg_causal_attn_bwd: unsupported head dims");
  PG_LAUNCH_CHECK("pg_causal_attn_bwd(dkv)");
  return 0;
}
