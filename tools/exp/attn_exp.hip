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
    default: return -1;
  }
  return (int)hipGetLastError();
}
