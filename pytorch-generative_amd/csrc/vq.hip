// vq.hip — vector quantisation (reference nn/utils.py:53-96) and the MSE loss of
// the VQ-VAE recipes (models/vae/vq_vae.py:127-136).
//
// STATUS: written at the end of round 2 after the round's GPU budget was spent — compiled for gfx950,
// NOT yet run on hardware. Nothing on the measured path calls these entry points; they are reached only
// through pytorch_generative_amd/experimental/vq.py, whose GPU tests are opt-in (PG_TEST_F4=1).
// The oracle they will be checked against is pinned already (oracle.ops.vector_quantize vs reference
// outputs, tests/golden/vq_*.pt).
//
// Work shape: N*H*W positions x K codes x D dims of multiply-add with K*D <= 32768 and at most a few
// 10^4 positions (8x8 .. 32x32 latent maps): HBM / launch bound byte work, not a GEMM worth the matrix
// pipe. One thread owns one position (its D values in registers, read coalesced from the NCHW planes),
// the codebook streams through LDS in chunks (every lane reads the same word: broadcast), and the
// distance is evaluated in the reference's form (|x|^2 + |e|^2) - 2 x.e with strict '<' so that the
// FIRST minimum wins like torch.argmin (nn/utils.py:61-68).
#include "common.h"

namespace {

constexpr int VQ_THREADS = 256;
constexpr int VQ_LDS_FLOATS = 8192;  // codes per chunk = VQ_LDS_FLOATS / DT

template <int DT>
__global__ void __launch_bounds__(VQ_THREADS) vq_assign_kernel(
    const float* __restrict__ x, const float* __restrict__ emb, int* __restrict__ idx,
    float* __restrict__ q, float* __restrict__ st, float* __restrict__ loss, int N, int D, int L, int K,
    float inv_numel) {
  extern __shared__ float sm[];  // codes [KC][DT], then |e|^2 [KC]
  const int KC = VQ_LDS_FLOATS / DT;
  float* e2 = sm + KC * DT;
  const int tid = threadIdx.x;
  const long p = (long)blockIdx.x * VQ_THREADS + tid;
  const bool valid = p < (long)N * L;
  const int n = valid ? (int)(p / L) : 0;
  const int l = valid ? (int)(p - (long)n * L) : 0;
  const float* xp = x + ((size_t)n * D) * L + l;
  float xr[DT];
  float x2 = 0.f;
#pragma unroll
  for (int d = 0; d < DT; ++d) {
    xr[d] = (valid && d < D) ? xp[(size_t)d * L] : 0.f;
    x2 += xr[d] * xr[d];
  }
  float best = 3.0e38f;
  int bi = 0;
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kc = min(KC, K - k0);
    __syncthreads();  // the previous chunk is consumed
    for (int i = tid; i < kc * DT; i += VQ_THREADS) {
      const int c = i / DT, d = i - c * DT;
      sm[i] = d < D ? emb[(size_t)(k0 + c) * D + d] : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < kc; c += VQ_THREADS) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DT; ++d) s += sm[c * DT + d] * sm[c * DT + d];
      e2[c] = s;
    }
    __syncthreads();
    for (int c = 0; c < kc; ++c) {
      float dot = 0.f;
#pragma unroll
      for (int d = 0; d < DT; ++d) dot = fmaf(xr[d], sm[c * DT + d], dot);
      const float dist = (x2 + e2[c]) - 2.f * dot;
      if (dist < best) {
        best = dist;
        bi = k0 + c;
      }
    }
  }
  float s = 0.f;
  if (valid) {
    idx[p] = bi;
    const float* ep = emb + (size_t)bi * D;
    float* qp = q + ((size_t)n * D) * L + l;
    float* sp = st + ((size_t)n * D) * L + l;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      if (d < D) {
        const float qv = ep[d];
        qp[(size_t)d * L] = qv;
        sp[(size_t)d * L] = xr[d] + (qv - xr[d]);  // the straight-through VALUE of nn/utils.py:95
        const float df = xr[d] - qv;
        s += df * df;
      }
    }
  }
  // commitment loss mse(x, q): block sum, one atomic per block
  s = pg_wave_sum(s);
  __shared__ float part[VQ_THREADS / 64];
  if ((tid & 63) == 0) part[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < VQ_THREADS / 64; ++w) t += part[w];
    atomicAdd(loss, t * inv_numel);
  }
}

__global__ void vq_zero_ws_kernel(float* __restrict__ count, float* __restrict__ sum, long nc, long ns) {
  const long i = blockIdx.x * (long)VQ_THREADS + threadIdx.x;
  if (i < nc) count[i] = 0.f;
  else if (i < nc + ns) sum[i - nc] = 0.f;
}

// batch statistics of the EMA update (nn/utils.py:81-82): count[k] = |{p: idx[p] = k}|,
// sum[k][d] = sum over those positions of x[p][d]
__global__ void vq_ema_stats_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                    float* __restrict__ count, float* __restrict__ sum, int N, int D, int L) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long)N * L) return;
  const int n = (int)(p / L), l = (int)(p - (long)n * L);
  const int k = idx[p];
  atomicAdd(count + k, 1.f);
  const float* xp = x + ((size_t)n * D) * L + l;
  for (int d = 0; d < D; ++d) atomicAdd(sum + (size_t)k * D + d, xp[(size_t)d * L]);
}

// cluster_size = cluster_size * decay + count * (1 - decay); embedding_avg likewise with sum;
// embedding = embedding_avg / (cluster_size + 1e-5)   (nn/utils.py:83-90). One thread per code.
__global__ void vq_ema_update_kernel(float* __restrict__ cluster_size, float* __restrict__ embedding_avg,
                                     float* __restrict__ embedding, const float* __restrict__ count,
                                     const float* __restrict__ sum, int K, int D, float decay) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const float cs = cluster_size[k] * decay + count[k] * (1.f - decay);
  cluster_size[k] = cs;
  const float inv = 1.f / (cs + 1e-5f);
  for (int d = 0; d < D; ++d) {
    const size_t i = (size_t)k * D + d;
    const float ea = embedding_avg[i] * decay + sum[i] * (1.f - decay);
    embedding_avg[i] = ea;
    embedding[i] = ea * inv;
  }
}

// dx = d_st + g_loss * 2 (x - q) / numel: the straight-through gradient plus the commitment loss's
__global__ void vq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ q,
                              const float* __restrict__ d_st, const float* __restrict__ g_loss,
                              float* __restrict__ dx, size_t n, float two_inv_numel) {
  const float g = g_loss[0] * two_inv_numel;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = d_st[i] + g * (x[i] - q[i]);
}

// loss[0] += mean((a - b)^2)
__global__ void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                               float* __restrict__ loss, size_t n, float inv_n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = a[i] - b[i];
    s += d * d;
  }
  s = pg_wave_sum(s);
  __shared__ float part[VQ_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < VQ_THREADS / 64; ++w) t += part[w];
    atomicAdd(loss, t * inv_n);
  }
}

// da = g * 2 (a - b) / n   (and db = -da when requested)
__global__ void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                               const float* __restrict__ g_loss, float* __restrict__ da,
                               float* __restrict__ db, size_t n, float two_inv_n) {
  const float g = g_loss[0] * two_inv_n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = g * (a[i] - b[i]);
    if (da) da[i] = v;
    if (db) db[i] = -v;
  }
}

inline int vq_blocks(size_t n) {
  const size_t b = (n + VQ_THREADS - 1) / VQ_THREADS;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

PG_EXPORT int pg_vq_assign(const float* x, const float* embedding, int* idx, float* q, float* st,
                           float* loss, int N, int D, int L, int K, void* stream) {
  PG_REQUIRE(x && embedding && idx && q && st && loss, PG_EINVAL, "pg_vq_assign: null pointer");
  PG_REQUIRE(N > 0 && L > 0 && K > 0, PG_EINVAL, "pg_vq_assign: non-positive dimension");
  PG_REQUIRE(D >= 1 && D <= 64, PG_ESHAPE, "pg_vq_assign: embedding_dim=%d not in [1,64]", D);
  hipStream_t s = (hipStream_t)stream;
  const long P = (long)N * L;
  const dim3 grid((unsigned)((P + VQ_THREADS - 1) / VQ_THREADS)), block(VQ_THREADS);
  const float inv_numel = 1.f / ((float)P * (float)D);
#define PG_VQ(DTV)                                                                                  \
  {                                                                                                 \
    const size_t shm = (size_t)(VQ_LDS_FLOATS + VQ_LDS_FLOATS / DTV) * sizeof(float);               \
    hipLaunchKernelGGL((vq_assign_kernel<DTV>), grid, block, shm, s, x, embedding, idx, q, st, loss, \
                       N, D, L, K, inv_numel);                                                      \
  }
  if (D <= 4) PG_VQ(4)
  else if (D <= 8) PG_VQ(8)
  else if (D <= 16) PG_VQ(16)
  else if (D <= 32) PG_VQ(32)
  else PG_VQ(64)
#undef PG_VQ
  PG_LAUNCH_CHECK("pg_vq_assign");
  return 0;
}

PG_EXPORT int pg_vq_ema_update(const float* x, const int* idx, float* cluster_size, float* embedding_avg,
                               float* embedding, float* count_ws, float* sum_ws, int N, int D, int L,
                               int K, float decay, void* stream) {
  PG_REQUIRE(x && idx && cluster_size && embedding_avg && embedding && count_ws && sum_ws, PG_EINVAL,
             "pg_vq_ema_update: null pointer");
  PG_REQUIRE(N > 0 && L > 0 && K > 0 && D > 0, PG_EINVAL, "pg_vq_ema_update: non-positive dimension");
  hipStream_t s = (hipStream_t)stream;
  // the two workspaces are zeroed by a kernel, not by memset nodes (attention_k4.hip, attn_delta_k4_kernel: a memset
  // node inside a replayed step graph went wrong on this ROCm)
  const long nz = (long)K * (D + 1);
  hipLaunchKernelGGL(vq_zero_ws_kernel, dim3((unsigned)((nz + VQ_THREADS - 1) / VQ_THREADS)), dim3(VQ_THREADS), 0, s,
                     count_ws, sum_ws, (long)K, (long)K * D);
  PG_LAUNCH_CHECK("pg_vq_ema_update(zero)");
  const long P = (long)N * L;
  hipLaunchKernelGGL(vq_ema_stats_kernel, dim3((unsigned)((P + VQ_THREADS - 1) / VQ_THREADS)),
                     dim3(VQ_THREADS), 0, s, x, idx, count_ws, sum_ws, N, D, L);
  PG_LAUNCH_CHECK("pg_vq_ema_update(stats)");
  hipLaunchKernelGGL(vq_ema_update_kernel, dim3((unsigned)((K + VQ_THREADS - 1) / VQ_THREADS)),
                     dim3(VQ_THREADS), 0, s, cluster_size, embedding_avg, embedding, count_ws, sum_ws, K, D,
                     decay);
  PG_LAUNCH_CHECK("pg_vq_ema_update");
  return 0;
}

PG_EXPORT int pg_vq_bwd(const float* x, const float* q, const float* d_st, const float* g_loss, float* dx,
                        size_t n, void* stream) {
  PG_REQUIRE(x && q && d_st && g_loss && dx, PG_EINVAL, "pg_vq_bwd: null pointer");
  PG_REQUIRE(n > 0, PG_EINVAL, "pg_vq_bwd: empty tensor");
  hipLaunchKernelGGL(vq_bwd_kernel, dim3(vq_blocks(n)), dim3(VQ_THREADS), 0, (hipStream_t)stream, x, q,
                     d_st, g_loss, dx, n, 2.f / (float)n);
  PG_LAUNCH_CHECK("pg_vq_bwd");
  return 0;
}

PG_EXPORT int pg_mse_fwd(const float* a, const float* b, float* loss, size_t n, void* stream) {
  PG_REQUIRE(a && b && loss, PG_EINVAL, "pg_mse_fwd: null pointer");
  PG_REQUIRE(n > 0, PG_EINVAL, "pg_mse_fwd: empty tensor");
  int blocks = vq_blocks(n);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(blocks), dim3(VQ_THREADS), 0, (hipStream_t)stream, a, b, loss, n,
                     1.f / (float)n);
  PG_LAUNCH_CHECK("pg_mse_fwd");
  return 0;
}

PG_EXPORT int pg_mse_bwd(const float* a, const float* b, const float* g_loss, float* da, float* db,
                         size_t n, void* stream) {
  PG_REQUIRE(a && b && g_loss && (da || db), PG_EINVAL, "pg_mse_bwd: null pointer");
  PG_REQUIRE(n > 0, PG_EINVAL, "pg_mse_bwd: empty tensor");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(vq_blocks(n)), dim3(VQ_THREADS), 0, (hipStream_t)stream, a, b,
                     g_loss, da, db, n, 2.f / (float)n);
  PG_LAUNCH_CHECK("pg_mse_bwd");
  return 0;
}
