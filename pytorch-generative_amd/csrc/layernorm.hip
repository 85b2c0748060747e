// layernorm.hip — LayerNorm over the channel axis computed DIRECTLY on NCHW data.
//
// Reference: NCHWLayerNorm.forward nn/convolution.py:72-75 does permute(0,2,3,1) ->
// nn.LayerNorm(C) -> permute(0,3,1,2) (two transposing copies + a channels-last view that
// then propagates through the model). Here one lane owns one pixel and walks the C channel
// planes: every load/store is a fully coalesced 256 B row of the (n, c) plane, no permutes.
// gamma/beta are wave-uniform -> scalar loads.
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_MAXC_REG = 64;  // channels kept in registers in the fast path

template <int CREG>
__global__ void __launch_bounds__(LN_THREADS)
ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
              const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
              float* __restrict__ rstd, int N, int C, int L, float eps) {
  const long p = (long)blockIdx.x * LN_THREADS + threadIdx.x;
  const long total = (long)N * L;
  if (p >= total) return;
  const int n = (int)(p / L);
  const int l = (int)(p - (long)n * L);
  const float* xp = x + (size_t)n * C * L + l;
  float* yp = y + (size_t)n * C * L + l;
  const float invC = 1.f / (float)C;
  float mu, rs;
  if (CREG > 0) {
    float v[CREG > 0 ? CREG : 1];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CREG; ++c) {
      v[c] = c < C ? xp[(size_t)c * L] : 0.f;
      s += v[c];
    }
    mu = s * invC;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CREG; ++c) {
      const float d = c < C ? v[c] - mu : 0.f;
      q = fmaf(d, d, q);
    }
    rs = rsqrtf(q * invC + eps);
    // rsqrtf is approximate on gfx950 (v_rsq_f32, 1 ulp): refine once for fp32 parity
    {
      const float a = q * invC + eps;
      rs = rs * (1.5f - 0.5f * a * rs * rs);
    }
#pragma unroll
    for (int c = 0; c < CREG; ++c)
      if (c < C) yp[(size_t)c * L] = fmaf((v[c] - mu) * rs, gamma[c], beta[c]);
  } else {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xp[(size_t)c * L];
    mu = s * invC;
    float q = 0.f;
    for (int c = 0; c < C; ++c) {
      const float d = xp[(size_t)c * L] - mu;
      q = fmaf(d, d, q);
    }
    const float a = q * invC + eps;
    rs = rsqrtf(a);
    rs = rs * (1.5f - 0.5f * a * rs * rs);
    for (int c = 0; c < C; ++c)
      yp[(size_t)c * L] = fmaf((xp[(size_t)c * L] - mu) * rs, gamma[c], beta[c]);
  }
  mean[p] = mu;
  rstd[p] = rs;
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g*xhat)),  g = dy*gamma
// dgamma[c] += sum_p dy*xhat ; dbeta[c] += sum_p dy
// One lane walks PPL pixels (stride = blockDim, so every access stays a coalesced row) keeping the
// pixel's C values of x and dy in registers (CREG > 0) and its dgamma/dbeta partials across pixels;
// one wave reduction + one atomic per channel per block at the end.
constexpr int LN_PPL = 4;

template <int CREG>
__global__ void __launch_bounds__(LN_THREADS)
ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
              const float* __restrict__ mean, const float* __restrict__ rstd,
              const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dgamma,
              float* __restrict__ dbeta, int N, int C, int L) {
  extern __shared__ float red[];  // [2][C][waves]
  constexpr int nw = LN_THREADS / 64;
  constexpr int CR = CREG > 0 ? CREG : 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long total = (long)N * L;
  const float invC = 1.f / (float)C;
  float pg[CR], pb[CR];
#pragma unroll
  for (int c = 0; c < CR; ++c) pg[c] = pb[c] = 0.f;

  for (int it = 0; it < LN_PPL; ++it) {
    const long p_raw = ((long)blockIdx.x * LN_PPL + it) * LN_THREADS + threadIdx.x;
    const bool live = p_raw < total;
    if (CREG > 0 && !live) break;      // register path: no cross-lane ops inside the pixel loop
    const long p = live ? p_raw : total - 1;
    const int n = (int)(p / L);
    const int l = (int)(p - (long)n * L);
    const size_t base = (size_t)n * C * L + l;
    const float mu = mean[p], rs = rstd[p];
    if (CREG > 0) {
      float xh[CR], dv[CR];
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int c = 0; c < CR; ++c) {
        const bool ok = c < C;
        xh[c] = ok ? (x[base + (size_t)c * L] - mu) * rs : 0.f;
        dv[c] = ok ? dy[base + (size_t)c * L] : 0.f;
        const float g = ok ? dv[c] * gamma[c] : 0.f;
        sg += g;
        sgx = fmaf(g, xh[c], sgx);
      }
      const float mg = sg * invC, mgx = sgx * invC;
#pragma unroll
      for (int c = 0; c < CR; ++c) {
        if (c < C) {
          dx[base + (size_t)c * L] = rs * (dv[c] * gamma[c] - mg - xh[c] * mgx);
          pg[c] = fmaf(dv[c], xh[c], pg[c]);
          pb[c] += dv[c];
        }
      }
    } else {
      float sg = 0.f, sgx = 0.f;
      for (int c = 0; c < C; ++c) {
        const float xh = (x[base + (size_t)c * L] - mu) * rs;
        const float g = dy[base + (size_t)c * L] * gamma[c];
        sg += g;
        sgx = fmaf(g, xh, sgx);
      }
      const float mg = sg * invC, mgx = sgx * invC;
      for (int c = 0; c < C; ++c) {
        const float xh = (x[base + (size_t)c * L] - mu) * rs;
        const float d = live ? dy[base + (size_t)c * L] : 0.f;
        if (live) dx[base + (size_t)c * L] = rs * (d * gamma[c] - mg - xh * mgx);
        // large C: reduce per channel straight away (no per-lane partial array)
        const float wg = pg_wave_sum(d * xh), wb = pg_wave_sum(d);
        if (lane == 0) {
          atomicAdd(&dgamma[c], wg);
          atomicAdd(&dbeta[c], wb);
        }
      }
    }
  }
  if (CREG > 0) {
#pragma unroll
    for (int c = 0; c < CR; ++c) {
      if (c < C) {
        const float wg = pg_wave_sum(pg[c]), wb = pg_wave_sum(pb[c]);
        if (lane == 0) {
          red[(0 * C + c) * nw + wave] = wg;
          red[(1 * C + c) * nw + wave] = wb;
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += LN_THREADS) {
      float s = 0.f;
      for (int w = 0; w < nw; ++w) s += red[i * nw + w];
      if (i < C) atomicAdd(&dgamma[i], s);
      else atomicAdd(&dbeta[i - C], s);
    }
  }
}

}  // namespace

PG_EXPORT int pg_nchw_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                                    float* y, float* mean, float* rstd, int N, int C, int L,
                                    float eps, void* stream) {
  PG_REQUIRE(x && gamma && beta && y && mean && rstd, PG_EINVAL, "pg_nchw_layernorm_fwd: null pointer");
  PG_REQUIRE(N > 0 && C > 0 && L > 0, PG_EINVAL, "pg_nchw_layernorm_fwd: bad dims");
  const long total = (long)N * L;
  dim3 grid((unsigned)((total + LN_THREADS - 1) / LN_THREADS));
  hipStream_t st = (hipStream_t)stream;
  if (C <= 16)
    hipLaunchKernelGGL(ln_fwd_kernel<16>, grid, dim3(LN_THREADS), 0, st, x, gamma, beta, y, mean, rstd, N, C, L, eps);
  else if (C <= LN_MAXC_REG)
    hipLaunchKernelGGL(ln_fwd_kernel<LN_MAXC_REG>, grid, dim3(LN_THREADS), 0, st, x, gamma, beta, y, mean, rstd, N, C, L, eps);
  else
    hipLaunchKernelGGL(ln_fwd_kernel<0>, grid, dim3(LN_THREADS), 0, st, x, gamma, beta, y, mean, rstd, N, C, L, eps);
  PG_LAUNCH_CHECK("pg_nchw_layernorm_fwd");
  return 0;
}

PG_EXPORT int pg_nchw_layernorm_bwd(const float* x, const float* gamma, const float* mean,
                                    const float* rstd, const float* dy, float* dx, float* dgamma,
                                    float* dbeta, int N, int C, int L, void* stream) {
  PG_REQUIRE(x && gamma && mean && rstd && dy && dx && dgamma && dbeta, PG_EINVAL,
             "pg_nchw_layernorm_bwd: null pointer");
  PG_REQUIRE(N > 0 && C > 0 && L > 0, PG_EINVAL, "pg_nchw_layernorm_bwd: bad dims");
  PG_REQUIRE(C <= 2048, PG_ESHAPE, "pg_nchw_layernorm_bwd: C=%d > 2048", C);
  const long total = (long)N * L;
  const long per_block = (long)LN_THREADS * LN_PPL;
  dim3 grid((unsigned)((total + per_block - 1) / per_block));
  const size_t shmem = (size_t)2 * C * (LN_THREADS / 64) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 16)
    hipLaunchKernelGGL(ln_bwd_kernel<16>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx, dgamma, dbeta, N, C, L);
  else if (C <= 32)
    hipLaunchKernelGGL(ln_bwd_kernel<32>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx, dgamma, dbeta, N, C, L);
  else if (C <= 64)
    hipLaunchKernelGGL(ln_bwd_kernel<64>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx, dgamma, dbeta, N, C, L);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<0>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx, dgamma, dbeta, N, C, L);
  PG_LAUNCH_CHECK("pg_nchw_layernorm_bwd");
  return 0;
}
