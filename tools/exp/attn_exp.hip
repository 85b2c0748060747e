// Experimental attention-forward variants for A/B timing (not part of libpg_hip.so).
#include <hip/hip_runtime.h>
#include <stdint.h>
#define NEG_BIG (-1.0e30f)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct A { const float* q; const float* k; const float* v; float* o; int N, heads, L; float scale2; };

template <int D, int CH>
__device__ __forceinline__ void loadrows(float (&dst)[D][CH], const float* base, int L, int pos) {
#pragma unroll
  for (int i = 0; i < D; ++i) { const float* rp = base + (size_t)i * L + pos;
#pragma unroll
    for (int c = 0; c < CH; ++c) dst[i][c] = rp[c]; }
}

// one query per lane; chunk CH; optional prefetch; unmasked-only timing variant (all keys < l0)
template <int CH, bool PREFETCH>
__global__ void __launch_bounds__(256) fwd_v1(const A a) {
  const int qb = gridDim.x - 1 - blockIdx.x; const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L; const int l0 = qb * 256 + wave * 64; if (l0 >= L) return;
  const int l = min(l0 + (int)(threadIdx.x & 63), L - 1);
  const size_t hb = ((size_t)n * a.heads + h) * 4 * L;
  const float* qp = a.q + hb; const float* kp = a.k + hb; const float* vp = a.v + hb;
  float qv[4]; for (int i = 0; i < 4; ++i) qv[i] = qp[(size_t)i * L + l] * a.scale2;
  float mrun = NEG_BIG, lsum = 0.f, acc[4] = {0, 0, 0, 0};
  const int m_end = ((min(l0 + 63, L - 1) + 1) / CH) * CH;  // ignore masking: timing only
  float kn[4][CH], vn[4][CH];
  if (PREFETCH) { loadrows<4, CH>(kn, kp, L, 0); loadrows<4, CH>(vn, vp, L, 0); }
  for (int m = 0; m < m_end; m += CH) {
    float kk[4][CH], vv[4][CH];
    if (PREFETCH) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < CH; ++c) { kk[i][c] = kn[i][c]; vv[i][c] = vn[i][c]; }
      const int mn = (m + CH < m_end) ? m + CH : m;
      loadrows<4, CH>(kn, kp, L, mn); loadrows<4, CH>(vn, vp, L, mn);
    } else { loadrows<4, CH>(kk, kp, L, m); loadrows<4, CH>(vv, vp, L, m); }
    float s[CH]; float cmax = NEG_BIG;
#pragma unroll
    for (int c = 0; c < CH; ++c) { float t = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) t = fmaf(qv[i], kk[i][c], t); s[c] = t; cmax = fmaxf(cmax, t); }
    const float mnew = fmaxf(mrun, cmax); const float alpha = fast_exp2(mrun - mnew);
    lsum *= alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= alpha;
#pragma unroll
    for (int c = 0; c < CH; ++c) { const float p = fast_exp2(s[c] - mnew); lsum += p;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(p, vv[j][c], acc[j]); }
    mrun = mnew;
  }
  float* op = a.o + hb + l; const float inv = 1.f / lsum;
  for (int j = 0; j < 4; ++j) op[(size_t)j * L] = acc[j] * inv;
}

// two queries per lane (wave covers 128 consecutive queries)
template <int CH, bool PREFETCH>
__global__ void __launch_bounds__(256) fwd_v2(const A a) {
  const int qb = gridDim.x - 1 - blockIdx.x; const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L; const int l0 = qb * 512 + wave * 128; if (l0 >= L) return;
  const int la = min(l0 + (int)(threadIdx.x & 63), L - 1), lb = min(la + 64, L - 1);
  const size_t hb = ((size_t)n * a.heads + h) * 4 * L;
  const float* qp = a.q + hb; const float* kp = a.k + hb; const float* vp = a.v + hb;
  float qa[4], qbv[4];
  for (int i = 0; i < 4; ++i) { qa[i] = qp[(size_t)i * L + la] * a.scale2; qbv[i] = qp[(size_t)i * L + lb] * a.scale2; }
  float ma = NEG_BIG, mb = NEG_BIG, lsa = 0.f, lsb = 0.f, aa[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
  const int m_end = ((min(l0 + 127, L - 1) + 1) / CH) * CH;
  float kn[4][CH], vn[4][CH];
  if (PREFETCH) { loadrows<4, CH>(kn, kp, L, 0); loadrows<4, CH>(vn, vp, L, 0); }
  for (int m = 0; m < m_end; m += CH) {
    float kk[4][CH], vv[4][CH];
    if (PREFETCH) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < CH; ++c) { kk[i][c] = kn[i][c]; vv[i][c] = vn[i][c]; }
      const int mn = (m + CH < m_end) ? m + CH : m;
      loadrows<4, CH>(kn, kp, L, mn); loadrows<4, CH>(vn, vp, L, mn);
    } else { loadrows<4, CH>(kk, kp, L, m); loadrows<4, CH>(vv, vp, L, m); }
    float sa[CH], sb[CH]; float ca = NEG_BIG, cb = NEG_BIG;
#pragma unroll
    for (int c = 0; c < CH; ++c) { float t = 0.f, u = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { t = fmaf(qa[i], kk[i][c], t); u = fmaf(qbv[i], kk[i][c], u); }
      sa[c] = t; sb[c] = u; ca = fmaxf(ca, t); cb = fmaxf(cb, u); }
    const float na = fmaxf(ma, ca), nb = fmaxf(mb, cb);
    const float ala = fast_exp2(ma - na), alb = fast_exp2(mb - nb);
    lsa *= ala; lsb *= alb;
#pragma unroll
    for (int j = 0; j < 4; ++j) { aa[j] *= ala; ab[j] *= alb; }
#pragma unroll
    for (int c = 0; c < CH; ++c) { const float p = fast_exp2(sa[c] - na), r = fast_exp2(sb[c] - nb); lsa += p; lsb += r;
#pragma unroll
      for (int j = 0; j < 4; ++j) { aa[j] = fmaf(p, vv[j][c], aa[j]); ab[j] = fmaf(r, vv[j][c], ab[j]); } }
    ma = na; mb = nb;
  }
  float* op = a.o + hb;
  for (int j = 0; j < 4; ++j) { op[(size_t)j * L + la] = aa[j] / lsa; op[(size_t)j * L + lb] = ab[j] / lsb; }
}


// v3: K/V of one (n,h) staged once in LDS as [m][8] (k0..3, v0..3); 2 queries per lane; keys broadcast from LDS
template <int QPL>
__global__ void __launch_bounds__(512) fwd_v3(const A a) {
  extern __shared__ float4 kvl[];  // [L][2]
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L;
  const size_t hb = ((size_t)n * a.heads + h) * 4 * L;
  const float* qp = a.q + hb; const float* kp = a.k + hb; const float* vp = a.v + hb;
  for (int m = threadIdx.x; m < L; m += blockDim.x) {
    kvl[2 * m] = make_float4(kp[m], kp[L + m], kp[2 * L + m], kp[3 * L + m]);
    kvl[2 * m + 1] = make_float4(vp[m], vp[L + m], vp[2 * L + m], vp[3 * L + m]);
  }
  __syncthreads();
  const int l0 = wave * 64 * QPL; if (l0 >= L) return;
  int lq[QPL]; float qv[QPL][4], mr[QPL], ls[QPL], acc[QPL][4];
#pragma unroll
  for (int u = 0; u < QPL; ++u) { lq[u] = min(l0 + u * 64 + (int)(threadIdx.x & 63), L - 1); mr[u] = NEG_BIG; ls[u] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { qv[u][i] = qp[(size_t)i * L + lq[u]] * a.scale2; acc[u][i] = 0.f; } }
  const int m_end = ((min(l0 + 64 * QPL - 1, L - 1) + 1) / 8) * 8;
  for (int m = 0; m < m_end; m += 8) {
    float4 kk[8], vv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { kk[c] = kvl[2 * (m + c)]; vv[c] = kvl[2 * (m + c) + 1]; }
#pragma unroll
    for (int u = 0; u < QPL; ++u) {
      float s[8]; float cmax = NEG_BIG;
#pragma unroll
      for (int c = 0; c < 8; ++c) { float t = qv[u][0] * kk[c].x; t = fmaf(qv[u][1], kk[c].y, t); t = fmaf(qv[u][2], kk[c].z, t); t = fmaf(qv[u][3], kk[c].w, t); s[c] = t; cmax = fmaxf(cmax, t); }
      const float mnew = fmaxf(mr[u], cmax); const float alpha = fast_exp2(mr[u] - mnew);
      ls[u] *= alpha;
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[u][j] *= alpha;
#pragma unroll
      for (int c = 0; c < 8; ++c) { const float p = fast_exp2(s[c] - mnew); ls[u] += p;
        acc[u][0] = fmaf(p, vv[c].x, acc[u][0]); acc[u][1] = fmaf(p, vv[c].y, acc[u][1]); acc[u][2] = fmaf(p, vv[c].z, acc[u][2]); acc[u][3] = fmaf(p, vv[c].w, acc[u][3]); }
      mr[u] = mnew;
    }
  }
#pragma unroll
  for (int u = 0; u < QPL; ++u) { float* op = a.o + hb + lq[u]; const float inv = 1.f / ls[u];
    for (int j = 0; j < 4; ++j) op[(size_t)j * L] = acc[u][j] * inv; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// v4: MFMA hybrid. S^T = K_tile(16x4) . Q^T(4x16) on v_mfma_f32_16x16x4_f32; lane = (query l&15 of each
// q-tile, keys 4*(l>>4)+r); softmax + PV on VALU with lane-local running max (lazy rescale), merged
// across the 4 lane groups at the end. Unmasked timing variant (keys < wave's l0 + 128).
template <int QT>
__global__ void __launch_bounds__(512) fwd_v4(const A a) {
  extern __shared__ float4 kvl[];
  float* kt = reinterpret_cast<float*>(kvl);        // K^T [4][Lp]
  const int L = a.L; const int Lp = ((L + 31) / 32) * 32 + 16;
  float4* vr = reinterpret_cast<float4*>(kt + 4 * Lp);  // V rows [L] float4
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const size_t hb = ((size_t)n * a.heads + h) * 4 * L;
  const float* qp = a.q + hb; const float* kp = a.k + hb; const float* vp = a.v + hb;
  for (int m = threadIdx.x; m < L; m += blockDim.x) {
    kt[m] = kp[m]; kt[Lp + m] = kp[L + m]; kt[2 * Lp + m] = kp[2 * L + m]; kt[3 * Lp + m] = kp[3 * L + m];
    vr[m] = make_float4(vp[m], vp[L + m], vp[2 * L + m], vp[3 * L + m]);
  }
  __syncthreads();
  const int l0 = wave * 16 * QT; if (l0 >= L) return;
  const int qi = lane & 15, g = lane >> 4;
  float qf[QT], mr[QT], ls[QT], acc[QT][4];
#pragma unroll
  for (int t = 0; t < QT; ++t) { const int lq = min(l0 + 16 * t + qi, L - 1); qf[t] = qp[(size_t)g * L + lq] * a.scale2; mr[t] = NEG_BIG; ls[t] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0.f; }
  const int m_end = ((min(l0 + 16 * QT - 1, L - 1) + 1) / 16) * 16;
  for (int m = 0; m < m_end; m += 16) {
    const float kf = kt[g * Lp + m + qi];                 // A operand: K[key m+qi][d=g]
    float4 vv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vv[r] = vr[m + 4 * g + r];  // V rows of this lane's 4 keys
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      f32x4 s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, qf[t], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float cmax = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
      if (__any(cmax > mr[t] + 8.f)) {      // lazy rescale (wave-uniform)
        const float mnew = fmaxf(mr[t], cmax); const float alpha = fast_exp2(mr[t] - mnew);
        ls[t] *= alpha; acc[t][0] *= alpha; acc[t][1] *= alpha; acc[t][2] *= alpha; acc[t][3] *= alpha; mr[t] = mnew;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float p = fast_exp2(s[r] - mr[t]); ls[t] += p;
        acc[t][0] = fmaf(p, vv[r].x, acc[t][0]); acc[t][1] = fmaf(p, vv[r].y, acc[t][1]); acc[t][2] = fmaf(p, vv[r].z, acc[t][2]); acc[t][3] = fmaf(p, vv[r].w, acc[t][3]); }
    }
  }
  // merge the 4 lane groups (xor 16, 32)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(mr[t], off, 64), lo = __shfl_xor(ls[t], off, 64);
      const float mn = fmaxf(mr[t], mo); const float ca = fast_exp2(mr[t] - mn), cb = fast_exp2(mo - mn);
      ls[t] = ls[t] * ca + lo * cb;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float oo = __shfl_xor(acc[t][j], off, 64); acc[t][j] = acc[t][j] * ca + oo * cb; }
      mr[t] = mn;
    }
    const int lq = l0 + 16 * t + qi;
    if (lq < L) { const float inv = 1.f / ls[t]; float* op = a.o + hb + lq;
      // group g writes channel g
      const float val = g == 0 ? acc[t][0] : (g == 1 ? acc[t][1] : (g == 2 ? acc[t][2] : acc[t][3]));
      op[(size_t)g * L] = val * inv; }
  }
}

// v5: v3 + balanced pairing: lane's two queries come from 64-blocks w and NB-1-w, so every wave of the
// (n,h) workgroup streams the same number of keys. Unmasked timing variant.
template <bool BOTH>
__device__ __forceinline__ void v5_chunk(const float4* kvl, int m, const float (&qv)[2][4], float (&mr)[2], float (&ls)[2], float (&acc)[2][4]) {
  float4 kk[8], vv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { kk[c] = kvl[2 * (m + c)]; vv[c] = kvl[2 * (m + c) + 1]; }
#pragma unroll
  for (int u = BOTH ? 0 : 1; u < 2; ++u) {
    float s[8]; float cmax = NEG_BIG;
#pragma unroll
    for (int c = 0; c < 8; ++c) { float t = qv[u][0] * kk[c].x; t = fmaf(qv[u][1], kk[c].y, t); t = fmaf(qv[u][2], kk[c].z, t); t = fmaf(qv[u][3], kk[c].w, t); s[c] = t; cmax = fmaxf(cmax, t); }
    const float mnew = fmaxf(mr[u], cmax); const float alpha = fast_exp2(mr[u] - mnew);
    ls[u] *= alpha;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[u][j] *= alpha;
#pragma unroll
    for (int c = 0; c < 8; ++c) { const float p = fast_exp2(s[c] - mnew); ls[u] += p;
      acc[u][0] = fmaf(p, vv[c].x, acc[u][0]); acc[u][1] = fmaf(p, vv[c].y, acc[u][1]); acc[u][2] = fmaf(p, vv[c].z, acc[u][2]); acc[u][3] = fmaf(p, vv[c].w, acc[u][3]); }
    mr[u] = mnew;
  }
}
__global__ void __launch_bounds__(512) fwd_v5(const A a) {
  extern __shared__ float4 kvl[];
  const int h = blockIdx.y, n = blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int L = a.L;
  const size_t hb = ((size_t)n * a.heads + h) * 4 * L;
  const float* qp = a.q + hb; const float* kp = a.k + hb; const float* vp = a.v + hb;
  for (int m = threadIdx.x; m < L; m += blockDim.x) {
    kvl[2 * m] = make_float4(kp[m], kp[L + m], kp[2 * L + m], kp[3 * L + m]);
    kvl[2 * m + 1] = make_float4(vp[m], vp[L + m], vp[2 * L + m], vp[3 * L + m]);
  }
  __syncthreads();
  const int NB = (L + 63) / 64;
  const int b1 = wave, b2 = NB - 1 - wave;
  if (b1 > b2) return;
  int lq[2] = {min(b1 * 64 + (int)(threadIdx.x & 63), L - 1), min(b2 * 64 + (int)(threadIdx.x & 63), L - 1)};
  float qv[2][4], mr[2] = {NEG_BIG, NEG_BIG}, ls[2] = {0.f, 0.f}, acc[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) { qv[u][i] = qp[(size_t)i * L + lq[u]] * a.scale2; acc[u][i] = 0.f; }
  const int e0 = (b1 == b2) ? 0 : ((min(b1 * 64 + 63, L - 1) + 1) / 8) * 8;
  const int e1 = ((min(b2 * 64 + 63, L - 1) + 1) / 8) * 8;
  int m = 0;
  for (; m < e0; m += 8) v5_chunk<true>(kvl, m, qv, mr, ls, acc);
  for (; m < e1; m += 8) v5_chunk<false>(kvl, m, qv, mr, ls, acc);
#pragma unroll
  for (int u = 0; u < 2; ++u) { if (u == 0 && b1 == b2) continue; float* op = a.o + hb + lq[u]; const float inv = 1.f / ls[u];
    for (int j = 0; j < 4; ++j) op[(size_t)j * L] = acc[u][j] * inv; }
}

extern "C" int exp_attn_fwd(int variant, const float* q, const float* k, const float* v, float* o, int N, int heads, int L, void* stream) {
  A a{q, k, v, o, N, heads, L, 1.44269504f * 0.5f};
  hipStream_t st = (hipStream_t)stream;
  dim3 g1((L + 255) / 256, heads, N), g2((L + 511) / 512, heads, N);
  switch (variant) {
    case 0: hipLaunchKernelGGL((fwd_v1<8, false>), g1, dim3(256), 0, st, a); break;
    case 1: hipLaunchKernelGGL((fwd_v1<4, false>), g1, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((fwd_v1<4, true>), g1, dim3(256), 0, st, a); break;
    case 3: hipLaunchKernelGGL((fwd_v2<4, false>), g2, dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((fwd_v2<4, true>), g2, dim3(256), 0, st, a); break;
    case 5: hipLaunchKernelGGL((fwd_v2<8, false>), g2, dim3(256), 0, st, a); break;
    case 6: hipLaunchKernelGGL((fwd_v1<2, true>), g1, dim3(256), 0, st, a); break;
    case 7: { int th = ((L + 127) / 128) * 64; hipLaunchKernelGGL((fwd_v3<2>), dim3(1, heads, N), dim3(th), L * 32, st, a); break; }
    case 8: { int th = ((L + 63) / 64) * 64; hipLaunchKernelGGL((fwd_v3<1>), dim3(1, heads, N), dim3(th > 1024 ? 1024 : th), L * 32, st, a); break; }
    case 9: { int th = ((L + 255) / 256) * 64; hipLaunchKernelGGL((fwd_v3<4>), dim3(1, heads, N), dim3(th), L * 32, st, a); break; }
    case 10: { int th = ((L + 127) / 128) * 64; int Lp = ((L + 31) / 32) * 32 + 16; hipLaunchKernelGGL((fwd_v4<8>), dim3(1, heads, N), dim3(th), (4 * Lp + 4 * L) * 4, st, a); break; }
    case 11: { int th = ((L + 63) / 64) * 64; int Lp = ((L + 31) / 32) * 32 + 16; hipLaunchKernelGGL((fwd_v4<4>), dim3(1, heads, N), dim3(th > 512 ? 512 : th), (4 * Lp + 4 * L) * 4, st, a); break; }
    case 12: { int NB = (L + 63) / 64; int th = ((NB + 1) / 2) * 64; hipLaunchKernelGGL(fwd_v5, dim3(1, heads, N), dim3(th), L * 32, st, a); break; }
    default: return -1;
  }
  return (int)hipGetLastError();
}
