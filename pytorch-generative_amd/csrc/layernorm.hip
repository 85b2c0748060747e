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
// One lane owns LN_PPL pixels (all their loads are issued before any arithmetic: the kernel is
// pure streaming) with the C values of x and dy in registers; dgamma/dbeta partials are reduced
// per wave (DPP), per block (LDS) and written as ONE partial row per block; a second tiny kernel
// sums the rows (same-address fp32 atomics cost ~65-100 ns per link on MI355X — measured).
constexpr int LN_PPL = 2;

template <int CREG>
__global__ void __launch_bounds__(LN_THREADS)
ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
              const float* __restrict__ mean, const float* __restrict__ rstd,
              const float* __restrict__ dy, const float* __restrict__ dx_add,
              float* __restrict__ dx, float* __restrict__ part, int N, int C, int L) {
  extern __shared__ float red[];  // [2][C][waves]
  constexpr int nw = LN_THREADS / 64;
  constexpr int CR = CREG > 0 ? CREG : 1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long total = (long)N * L;
  const float invC = 1.f / (float)C;

  if (CREG > 0) {
    float xv[LN_PPL][CR], dv[LN_PPL][CR], av[LN_PPL][CR], mu[LN_PPL], rs[LN_PPL];
    size_t base[LN_PPL];
    bool live[LN_PPL];
#pragma unroll
    for (int it = 0; it < LN_PPL; ++it) {
      const long p_raw = ((long)blockIdx.x * LN_PPL + it) * LN_THREADS + threadIdx.x;
      live[it] = p_raw < total;
      const long p = live[it] ? p_raw : total - 1;
      const int n = (int)(p / L);
      base[it] = (size_t)n * C * L + (size_t)(p - (long)n * L);
      mu[it] = mean[p];
      rs[it] = rstd[p];
#pragma unroll
      for (int c = 0; c < CR; ++c) {
        xv[it][c] = c < C ? x[base[it] + (size_t)c * L] : 0.f;
        dv[it][c] = (c < C && live[it]) ? dy[base[it] + (size_t)c * L] : 0.f;
        av[it][c] = (dx_add != nullptr && c < C) ? dx_add[base[it] + (size_t)c * L] : 0.f;  // skip-path gradient
      }
    }
    float pg[CR], pb[CR];
#pragma unroll
    for (int c = 0; c < CR; ++c) pg[c] = pb[c] = 0.f;
#pragma unroll
    for (int it = 0; it < LN_PPL; ++it) {
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int c = 0; c < CR; ++c) {
        xv[it][c] = c < C ? (xv[it][c] - mu[it]) * rs[it] : 0.f;  // xhat
        const float g = c < C ? dv[it][c] * gamma[c] : 0.f;
        sg += g;
        sgx = fmaf(g, xv[it][c], sgx);
      }
      const float mg = sg * invC, mgx = sgx * invC;
#pragma unroll
      for (int c = 0; c < CR; ++c) {
        if (c < C) {
          if (live[it])
            dx[base[it] + (size_t)c * L] = rs[it] * (dv[it][c] * gamma[c] - mg - xv[it][c] * mgx) + av[it][c];
          pg[c] = fmaf(dv[it][c], xv[it][c], pg[c]);
          pb[c] += dv[it][c];
        }
      }
    }
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
  } else {
    // large C: stream the channels twice, reduce each channel straight away
    for (int c = threadIdx.x; c < 2 * C * nw; c += LN_THREADS) red[c] = 0.f;
    __syncthreads();
    for (int it = 0; it < LN_PPL; ++it) {
      const long p_raw = ((long)blockIdx.x * LN_PPL + it) * LN_THREADS + threadIdx.x;
      const bool live = p_raw < total;
      const long p = live ? p_raw : total - 1;
      const int n = (int)(p / L);
      const size_t base = (size_t)n * C * L + (size_t)(p - (long)n * L);
      const float mu = mean[p], rs = rstd[p];
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
        if (live)
          dx[base + (size_t)c * L] = rs * (d * gamma[c] - mg - xh * mgx) +
                                     (dx_add != nullptr ? dx_add[base + (size_t)c * L] : 0.f);
        const float wg = pg_wave_sum(d * xh), wb = pg_wave_sum(d);
        if (lane == 0) {
          red[(0 * C + c) * nw + wave] += wg;
          red[(1 * C + c) * nw + wave] += wb;
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += LN_THREADS) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += red[i * nw + w];
    part[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

// out[i] += sum_g part[g*n + i]: 8 slots x 32 row groups per block (short dependent chains).
__global__ void __launch_bounds__(256)
rows_reduce_add_kernel(const float* __restrict__ part, int G, int n, float* __restrict__ out0,
                       float* __restrict__ out1, int split) {
  __shared__ float red[32][9];
  const int sl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int s = blockIdx.x * 8 + sl;
  float a0 = 0.f, a1 = 0.f;
  if (s < n) {
    const float* p = part + s;
    int g = rg;
    for (; g + 32 < G; g += 64) {
      a0 += p[(size_t)g * n];
      a1 += p[(size_t)(g + 32) * n];
    }
    if (g < G) a0 += p[(size_t)g * n];
  }
  red[rg][sl] = a0 + a1;
  __syncthreads();
  if (rg != 0 || s >= n) return;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) acc += red[r][sl];
  if (s < split) out0[s] += acc;
  else out1[s - split] += acc;
}

}  // namespace

static long ln_bwd_blocks(int N, int L) {
  const long total = (long)N * L;
  const long per_block = (long)LN_THREADS * LN_PPL;
  return (total + per_block - 1) / per_block;
}

PG_EXPORT size_t pg_nchw_layernorm_bwd_workspace_floats(int N, int C, int L) {
  return (size_t)ln_bwd_blocks(N, L) * 2 * C;
}

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

namespace {
int ln_bwd_impl(const char* who, const float* x, const float* gamma, const float* mean, const float* rstd,
                const float* dy, const float* dx_add, float* dx, float* dgamma, float* dbeta, int N,
                int C, int L, float* workspace, size_t workspace_floats, void* stream) {
  PG_REQUIRE(x && gamma && mean && rstd && dy && dx && dgamma && dbeta && workspace, PG_EINVAL,
             "%s: null pointer", who);
  PG_REQUIRE(N > 0 && C > 0 && L > 0, PG_EINVAL, "%s: bad dims", who);
  PG_REQUIRE(C <= 2048, PG_ESHAPE, "%s: C=%d > 2048", who, C);
  PG_REQUIRE(workspace_floats >= pg_nchw_layernorm_bwd_workspace_floats(N, C, L), PG_EINVAL,
             "%s: workspace too small", who);
  const long blocks = ln_bwd_blocks(N, L);
  dim3 grid((unsigned)blocks);
  const size_t shmem = (size_t)2 * C * (LN_THREADS / 64) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 16)
    hipLaunchKernelGGL(ln_bwd_kernel<16>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx_add, dx, workspace, N, C, L);
  else if (C <= 32)
    hipLaunchKernelGGL(ln_bwd_kernel<32>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx_add, dx, workspace, N, C, L);
  else
    hipLaunchKernelGGL(ln_bwd_kernel<0>, grid, dim3(LN_THREADS), shmem, st, x, gamma, mean, rstd, dy, dx_add, dx, workspace, N, C, L);
  PG_LAUNCH_CHECK(who);
  hipLaunchKernelGGL(rows_reduce_add_kernel, dim3((unsigned)((2 * C + 7) / 8)), dim3(256), 0, st,
                     workspace, (int)blocks, 2 * C, dgamma, dbeta, C);
  PG_LAUNCH_CHECK(who);
  return 0;
}
}  // namespace

PG_EXPORT int pg_nchw_layernorm_bwd(const float* x, const float* gamma, const float* mean,
                                    const float* rstd, const float* dy, float* dx, float* dgamma,
                                    float* dbeta, int N, int C, int L, float* workspace,
                                    size_t workspace_floats, void* stream) {
  return ln_bwd_impl("pg_nchw_layernorm_bwd", x, gamma, mean, rstd, dy, nullptr, dx, dgamma, dbeta, N, C,
                     L, workspace, workspace_floats, stream);
}

PG_EXPORT int pg_nchw_layernorm_bwd_res(const float* x, const float* gamma, const float* mean,
                                        const float* rstd, const float* dy, const float* dx_add,
                                        float* dx, float* dgamma, float* dbeta, int N, int C, int L,
                                        float* workspace, size_t workspace_floats, void* stream) {
  PG_REQUIRE(dx_add, PG_EINVAL, "pg_nchw_layernorm_bwd_res: null pointer");
  return ln_bwd_impl("pg_nchw_layernorm_bwd_res", x, gamma, mean, rstd, dy, dx_add, dx, dgamma, dbeta, N,
                     C, L, workspace, workspace_floats, stream);
}
