// dmol.hip — discretized mixture-of-logistics negative log-likelihood (the PixelCNN++ loss named by
// BASELINE.json configs[2]), forward and backward.
//
// The reference repository has no implementation of this loss, so there is no reference call site to
// cite; the algorithm is the published one — Salimans, Karpathy, Chen, Kingma, "PixelCNN++", ICLR 2017,
// eq. (2) (logistic mixture discretized to 256 bins, the edge bins absorbing the tails) and eq. (3)
// (the mean of G depends linearly on R, the mean of B on R and G) — restated with its numerical
// safeguards in oracle/dmol.py, which tests/test_dmol_cpu.py pins by analytic known answers.
//
// Layout: network output l (N, 10 K, H, W) NCHW — channels [0, K) mixture logits, then for sub-pixel
// c = 0, 1, 2: K means, K log-scales (clamped at -7), K raw coefficients (tanh applied); images x
// (N, 3, H, W) in [-1, 1]. Lane = pixel: every channel read is a coalesced 256-byte row segment.
//   loss[0] += -(1 / N) sum_{n, pixels} logsumexp_k [ log_softmax(logits)_k + sum_c log P(x_c | k) ]
// HBM bound: 10 K + 3 floats read per pixel (backward: read again, 10 K written); K <= 16.
#include "common.h"

namespace {

constexpr int DM_THREADS = 256;
constexpr float LOG_SCALE_MIN = -7.0f;
constexpr float BIN = 1.0f / 255.0f;
constexpr float MASS_SWITCH = 1e-5f;
constexpr float MASS_FLOOR = 1e-12f;
constexpr float LOG_127_5 = 4.8481163885f;

__device__ __forceinline__ float softplusf_(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoidf2_(float z) { return 1.f / (1.f + expf(-z)); }

// log P(x | mean m, log-scale s) of one sub-pixel and its derivatives w.r.t. m and s (s already clamped;
// ds excludes the clamp's mask)
__device__ __forceinline__ float subpixel(float x, float m, float s, float& dm, float& ds) {
  const float inv_s = expf(-s), cen = x - m;
  const float pin = inv_s * (cen + BIN), nin = inv_s * (cen - BIN), mid = inv_s * cen;
  if (x < -0.999f) {  // value 0: everything below the first bin edge. log sigmoid(pin)
    const float d = 1.f - sigmoidf2_(pin);
    dm = -inv_s * d;
    ds = -pin * d;
    return pin - softplusf_(pin);
  }
  if (x > 0.999f) {   // value 255: everything above the last edge. log(1 - sigmoid(nin))
    const float d = -sigmoidf2_(nin);
    dm = -inv_s * d;
    ds = -nin * d;
    return -softplusf_(nin);
  }
  const float cp = sigmoidf2_(pin), cm = sigmoidf2_(nin), delta = cp - cm;
  if (delta > MASS_SWITCH) {
    const float dp = cp * (1.f - cp) / delta, dn = -cm * (1.f - cm) / delta;  // d / d pin, d / d nin
    dm = -inv_s * (dp + dn);
    ds = -(pin * dp + nin * dn);
    return logf(fmaxf(delta, MASS_FLOOR));
  }
  const float d = 1.f - 2.f * sigmoidf2_(mid);  // log-density at the bin centre - log 127.5
  dm = -inv_s * d;
  ds = -mid * d - 1.f;
  return mid - s - 2.f * softplusf_(mid) - LOG_127_5;
}

struct DmArgs {
  const float* l; const float* x; const float* gscale; float* loss; float* dl;
  long total;  // N * L pixels
  int K, L;
  float invN;
};

// joint log-probability of component k at this pixel (without the log_softmax normaliser); the
// derivative slots are filled when GRAD
template <bool GRAD>
__device__ __forceinline__ float component(const float* lp, int K, size_t L, int k, float xr, float xg, float xb,
                                           float (&d)[9]) {
  // channel of (sub-pixel c, field f, component k): K + c * 3K + f * K + k
  auto at = [&](int c, int f) { return lp[(size_t)(K + c * 3 * K + f * K + k) * L]; };
  const float c0 = tanhf(at(0, 2)), c1 = tanhf(at(1, 2)), c2 = tanhf(at(2, 2));
  const float s0r = at(0, 1), s1r = at(1, 1), s2r = at(2, 1);
  const float s0 = fmaxf(s0r, LOG_SCALE_MIN), s1 = fmaxf(s1r, LOG_SCALE_MIN), s2 = fmaxf(s2r, LOG_SCALE_MIN);
  const float m0 = at(0, 0), m1 = at(1, 0) + c0 * xr, m2 = at(2, 0) + c1 * xr + c2 * xg;
  float dm0, ds0, dm1, ds1, dm2, ds2;
  const float lp0 = subpixel(xr, m0, s0, dm0, ds0);
  const float lp1 = subpixel(xg, m1, s1, dm1, ds1);
  const float lp2 = subpixel(xb, m2, s2, dm2, ds2);
  if (GRAD) {
    d[0] = dm0; d[1] = s0r > LOG_SCALE_MIN ? ds0 : 0.f; d[2] = dm1 * xr * (1.f - c0 * c0);   // coefficient 0: G on R
    d[3] = dm1; d[4] = s1r > LOG_SCALE_MIN ? ds1 : 0.f; d[5] = dm2 * xr * (1.f - c1 * c1);   // coefficient 1: B on R
    d[6] = dm2; d[7] = s2r > LOG_SCALE_MIN ? ds2 : 0.f; d[8] = dm2 * xg * (1.f - c2 * c2);   // coefficient 2: B on G
  }
  return lp0 + lp1 + lp2;
}

template <bool GRAD>
__global__ void __launch_bounds__(DM_THREADS) dmol_kernel(const DmArgs a) {
  const long stride = (long)gridDim.x * blockDim.x;
  const int K = a.K;
  const size_t L = (size_t)a.L;
  float acc = 0.f;
  const float g = GRAD ? a.gscale[0] * a.invN : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.total; i += stride) {
    const long n = i / a.L;
    const int p = (int)(i - n * a.L);
    const float* lp = a.l + (size_t)n * 10 * K * L + p;
    const float* xp = a.x + (size_t)n * 3 * L + p;
    const float xr = xp[0], xg = xp[L], xb = xp[2 * L];
    // log_softmax normaliser of the mixture logits
    float mx = -1e30f;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, lp[(size_t)k * L]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(lp[(size_t)k * L] - mx);
    const float lse_logits = mx + logf(se);
    // logsumexp over the components (online)
    float m = -1e30f, s = 0.f;
    float d[9];
    for (int k = 0; k < K; ++k) {
      const float v = lp[(size_t)k * L] - lse_logits + component<false>(lp, K, L, k, xr, xg, xb, d);
      const float nm = fmaxf(m, v);
      s = s * expf(m - nm) + expf(v - nm);
      m = nm;
    }
    const float ll = m + logf(s);
    if (!GRAD) {
      acc -= ll;
    } else {
      float* dp = a.dl + (size_t)n * 10 * K * L + p;
      for (int k = 0; k < K; ++k) {
        const float logit = lp[(size_t)k * L];
        const float v = logit - lse_logits + component<true>(lp, K, L, k, xr, xg, xb, d);
        const float w = expf(v - ll);                  // posterior of component k
        const float prior = expf(logit - lse_logits);  // softmax(logits)_k
        dp[(size_t)k * L] = -g * (w - prior);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dp[(size_t)(K + c * 3 * K + 0 * K + k) * L] = -g * w * d[3 * c + 0];
          dp[(size_t)(K + c * 3 * K + 1 * K + k) * L] = -g * w * d[3 * c + 1];
        }
        // raw coefficient c lives in the third field of sub-pixel c: 0 couples G to R, 1 B to R, 2 B to G
        dp[(size_t)(K + 0 * 3 * K + 2 * K + k) * L] = -g * w * d[2];
        dp[(size_t)(K + 1 * 3 * K + 2 * K + k) * L] = -g * w * d[5];
        dp[(size_t)(K + 2 * 3 * K + 2 * K + k) * L] = -g * w * d[8];
      }
    }
  }
  if (!GRAD) {
    acc = pg_wave_sum(acc);
    __shared__ float part[DM_THREADS / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < DM_THREADS / 64; ++w) t += part[w];
      atomicAdd(a.loss, t * a.invN);
    }
  }
}

int dm_check(const char* who, int N, int K, int L) {
  PG_REQUIRE(N > 0 && L > 0, PG_EINVAL, "%s: non-positive dimension", who);
  PG_REQUIRE(K >= 1 && K <= 16, PG_ESHAPE, "%s: %d mixture components not in [1, 16]", who, K);
  return 0;
}

}  // namespace

PG_EXPORT int pg_dmol_fwd(const float* l, const float* x, float* loss, int N, int K, int L, void* stream) {
  PG_REQUIRE(l && x && loss, PG_EINVAL, "pg_dmol_fwd: null pointer");
  if (int rc = dm_check("pg_dmol_fwd", N, K, L)) return rc;
  DmArgs a = {};
  a.l = l; a.x = x; a.loss = loss; a.total = (long)N * L; a.K = K; a.L = L; a.invN = 1.f / (float)N;
  long blocks = (a.total + DM_THREADS - 1) / DM_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(dmol_kernel<false>, dim3((unsigned)blocks), dim3(DM_THREADS), 0, (hipStream_t)stream, a);
  PG_LAUNCH_CHECK("pg_dmol_fwd");
  return 0;
}

PG_EXPORT int pg_dmol_bwd(const float* l, const float* x, const float* gscale, float* dl, int N, int K, int L,
                          void* stream) {
  PG_REQUIRE(l && x && gscale && dl, PG_EINVAL, "pg_dmol_bwd: null pointer");
  if (int rc = dm_check("pg_dmol_bwd", N, K, L)) return rc;
  DmArgs a = {};
  a.l = l; a.x = x; a.gscale = gscale; a.dl = dl; a.total = (long)N * L; a.K = K; a.L = L; a.invN = 1.f / (float)N;
  long blocks = (a.total + DM_THREADS - 1) / DM_THREADS;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dmol_kernel<true>, dim3((unsigned)blocks), dim3(DM_THREADS), 0, (hipStream_t)stream, a);
  PG_LAUNCH_CHECK("pg_dmol_bwd");
  return 0;
}
