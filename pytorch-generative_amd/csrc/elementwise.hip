// elementwise.hip — the memory-bound pieces of the hot path (fp32, float4 grid-stride).
//
// Reference call sites: ReLU/ELU/GELU (pixel_cnn.py:33-50, pixel_snail.py:27-28,
// image_gpt.py:44), GatedActivation (nn/convolution.py:62-66), residual adds
// (image_gpt.py:50-52,107-108), the learned positional map (image_gpt.py:86,106),
// `weight.data *= mask` (nn/convolution.py:42), image_positional_encoding
// (nn/attention.py:37-57) and the BCE-with-logits loss (image_gpt.py:158-162).
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;

inline int ew_blocks(size_t n_items) {
  size_t b = (n_items + EW_THREADS - 1) / EW_THREADS;
  const size_t cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond that
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int ACT>
__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                               int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const size_t n4 = n >> 2;
    for (; i < n4; i += stride) {
      float4 v = reinterpret_cast<const float4*>(x)[i];
      v.x = pg_apply_act(v.x, ACT); v.y = pg_apply_act(v.y, ACT);
      v.z = pg_apply_act(v.z, ACT); v.w = pg_apply_act(v.w, ACT);
      reinterpret_cast<float4*>(y)[i] = v;
    }
    // tail
    for (size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
      y[t] = pg_apply_act(x[t], ACT);
  } else {
    for (; i < n; i += stride) y[i] = pg_apply_act(x[i], ACT);
  }
}

template <int ACT>
__global__ void act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                               float* __restrict__ dx, size_t n, int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const size_t n4 = n >> 2;
    for (; i < n4; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      float4 g = reinterpret_cast<const float4*>(dy)[i];
      g.x *= pg_act_grad(v.x, ACT); g.y *= pg_act_grad(v.y, ACT);
      g.z *= pg_act_grad(v.z, ACT); g.w *= pg_act_grad(v.w, ACT);
      reinterpret_cast<float4*>(dx)[i] = g;
    }
    for (size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
      dx[t] = dy[t] * pg_act_grad(x[t], ACT);
  } else {
    for (; i < n; i += stride) dx[i] = dy[i] * pg_act_grad(x[i], ACT);
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// x: (N, 2C, L) -> y: (N, C, L); item = (n, c, l)
__global__ void gated_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                 size_t CL, size_t total, int gate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t n = i / CL;
    const size_t r = i - n * CL;
    const float a = x[n * 2 * CL + r];
    const float b = x[n * 2 * CL + CL + r];
    const float f = gate == PG_GATE_TANH ? tanhf(a) : a;
    y[i] = f * sigmoidf_(b);
  }
}

__global__ void gated_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                 float* __restrict__ dx, int C, size_t CL, size_t total,
                                 int gate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t n = i / CL;
    const size_t r = i - n * CL;
    const float a = x[n * 2 * CL + r];
    const float b = x[n * 2 * CL + CL + r];
    const float g = dy[i];
    const float s = sigmoidf_(b);
    float f, df;
    if (gate == PG_GATE_TANH) {
      f = tanhf(a);
      df = 1.f - f * f;
    } else {
      f = a;
      df = 1.f;
    }
    dx[n * 2 * CL + r] = g * df * s;
    dx[n * 2 * CL + CL + r] = g * f * s * (1.f - s);
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                           float* __restrict__ out, size_t n, int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const size_t n4 = n >> 2;
    for (; i < n4; i += stride) {
      const float4 u = reinterpret_cast<const float4*>(a)[i];
      const float4 v = reinterpret_cast<const float4*>(b)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    for (size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
      out[t] = a[t] + b[t];
  } else {
    for (; i < n; i += stride) out[i] = a[i] + b[i];
  }
}

__global__ void fill_kernel(float* __restrict__ out, float value, size_t n, int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec) {
    const size_t n4 = n >> 2;
    const float4 v = make_float4(value, value, value, value);
    for (; i < n4; i += stride) reinterpret_cast<float4*>(out)[i] = v;
    for (size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) out[t] = value;
  } else {
    for (; i < n; i += stride) out[i] = value;
  }
}

struct SumRowsArgs {
  const float* rows[32];
  long bs[32];
  float* out;
  long per;
  size_t n;
  int n_rows, dense;
};

__global__ void sum_rows_kernel(const SumRowsArgs a) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
    float s = 0.f;
    if (a.dense) {
      for (int k = 0; k < a.n_rows; ++k) s += a.rows[k][i];  // fixed order: bit-reproducible
    } else {
      const long b = (long)(i / (size_t)a.per), r = (long)(i - (size_t)b * a.per);
      for (int k = 0; k < a.n_rows; ++k) s += a.rows[k][b * a.bs[k] + r];
    }
    a.out[i] = s;
  }
}

__global__ void mul_inplace_kernel(float* __restrict__ w, const float* __restrict__ m, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] *= m[i];
}

__global__ void add_bcast_fwd_kernel(const float* __restrict__ x, const float* __restrict__ p,
                                     float* __restrict__ y, size_t per, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    y[i] = x[i] + p[i % per];
}

// dp[i] += sum_n dy[n, i]. Block = 32 columns x 32 row groups: a thread sums every 32nd row of its
// column with 8 loads in flight, the row groups meet in LDS (deterministic; the first version — one
// thread per column walking all N rows — took 238 us for the (1024, 784) positional-embedding grad).
__global__ void __launch_bounds__(1024) add_bcast_bwd_kernel(const float* __restrict__ dy,
                                                             float* __restrict__ dp, int N, size_t per) {
  __shared__ float red[32][33];
  const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const size_t i = (size_t)blockIdx.x * 32 + col;
  float s = 0.f;
  if (i < per) {
    int n = rg;
    for (; n + 7 * 32 < N; n += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = dy[(size_t)(n + 32 * u) * per + i];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; n < N; n += 32) s += dy[(size_t)n * per + i];
  }
  red[rg][col] = s;
  __syncthreads();
  if (rg == 0 && i < per) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) t += red[r][col];
    dp[i] += t;
  }
}

// torch.arange(-0.5, 0.5, 1/h) on torch-CPU (the reference builds the encoding on the host,
// nn/attention.py:53-54) is NOT simply (float)(start + i*step): ATen's vectorised loop handles
// the first floor(n/16)*16 elements as 8-wide vectors, each computed as
//   (float) fma(j, step, (double)(float) fma(b0, step, start)),  b0 = 8*(i/8), j = i - b0
// and the remaining < 16 elements as (float) fma(i, step, start), all in double with fused
// multiply-adds (verified bit-exact for every h in 2..600 on this image's AVX-512 torch build).
__device__ __forceinline__ float arange_like_torch_cpu(int i, int n, double step) {
  const double start = -0.5;
  const int nvec = (n / 16) * 16;
  if (i < nvec) {
    const int b0 = (i / 8) * 8;
    const float base = (float)fma((double)b0, step, start);
    return (float)fma((double)(i - b0), step, (double)base);
  }
  return (float)fma((double)i, step, start);
}

__global__ void posenc_kernel(float* __restrict__ out, int N, int H, int W) {
  const size_t HW = (size_t)H * W;
  const size_t total = (size_t)N * 2 * HW;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t r = i % (2 * HW);
    const int ch = (int)(r / HW);
    const int hw = (int)(r - (size_t)ch * HW);
    const int row = hw / W, col = hw - row * W;
    out[i] = ch == 0 ? arange_like_torch_cpu(row, H, 1.0 / (double)H)
                     : arange_like_torch_cpu(col, W, 1.0 / (double)W);
  }
}

// loss[0] += (1/N) sum [max(z,0) - z*x + log1p(exp(-|z|))]
__global__ void bce_fwd_kernel(const float* __restrict__ z, const float* __restrict__ x,
                               float* __restrict__ loss, size_t total, float invN) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const float zi = z[i], xi = x[i];
    s += fmaxf(zi, 0.f) - zi * xi + log1pf(expf(-fabsf(zi)));
  }
  s = pg_wave_sum(s);
  __shared__ float part[EW_THREADS / 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) part[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < EW_THREADS / 64; ++w) t += part[w];
    atomicAdd(loss, t * invN);
  }
}

__global__ void bce_bwd_kernel(const float* __restrict__ z, const float* __restrict__ x,
                               const float* __restrict__ gscale, float* __restrict__ dz,
                               size_t total, float invN) {
  const float g = gscale[0] * invN;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride)
    dz[i] = g * (sigmoidf_(z[i]) - x[i]);
}

// dx = dy * act'(v), v = y - res: the activation's derivative recovered from its OUTPUT
// (ELU: v > 0 ? 1 : v + 1; ReLU: v > 0) — backward of a convolution epilogue `y = act(conv) + res`
template <int ACT>
__global__ void act_bwd_out_kernel(const float* __restrict__ y, const float* __restrict__ res,
                                   const float* __restrict__ dy, float* __restrict__ dx, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(y)[i];
    if (res) {
      const float4 r = reinterpret_cast<const float4*>(res)[i];
      v.x -= r.x; v.y -= r.y; v.z -= r.z; v.w -= r.w;
    }
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    g.x *= pg_act_grad_out(v.x, ACT); g.y *= pg_act_grad_out(v.y, ACT);
    g.z *= pg_act_grad_out(v.z, ACT); g.w *= pg_act_grad_out(v.w, ACT);
    reinterpret_cast<float4*>(dx)[i] = g;
  }
}

// float4 gate kernels (L % 4 == 0): item = (n, c, quad of pixels)
__global__ void gated_fwd4_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                  float* __restrict__ y, size_t CL4, size_t total4, int gate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const size_t n = i / CL4;
    const size_t r = i - n * CL4;
    const float4 a = reinterpret_cast<const float4*>(x)[n * 2 * CL4 + r];
    const float4 b = reinterpret_cast<const float4*>(x)[n * 2 * CL4 + CL4 + r];
    float4 o;
    if (gate == PG_GATE_TANH) {
      o.x = tanhf(a.x) * sigmoidf_(b.x); o.y = tanhf(a.y) * sigmoidf_(b.y);
      o.z = tanhf(a.z) * sigmoidf_(b.z); o.w = tanhf(a.w) * sigmoidf_(b.w);
    } else {
      o.x = a.x * sigmoidf_(b.x); o.y = a.y * sigmoidf_(b.y);
      o.z = a.z * sigmoidf_(b.z); o.w = a.w * sigmoidf_(b.w);
    }
    if (res) {
      const float4 rv = reinterpret_cast<const float4*>(res)[i];
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

__device__ __forceinline__ void gate_grad(float a, float b, float g, int gate, float& da, float& db) {
  const float sg = sigmoidf_(b);
  if (gate == PG_GATE_TANH) {
    const float t = tanhf(a);
    da = g * sg * (1.f - t * t);
    db = g * t * sg * (1.f - sg);
  } else {
    da = g * sg;
    db = g * a * sg * (1.f - sg);
  }
}

__global__ void gated_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                  float* __restrict__ dx, size_t CL4, size_t total4, int gate) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += stride) {
    const size_t n = i / CL4;
    const size_t r = i - n * CL4;
    const float4 a = reinterpret_cast<const float4*>(x)[n * 2 * CL4 + r];
    const float4 b = reinterpret_cast<const float4*>(x)[n * 2 * CL4 + CL4 + r];
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    float4 da, db;
    gate_grad(a.x, b.x, g.x, gate, da.x, db.x);
    gate_grad(a.y, b.y, g.y, gate, da.y, db.y);
    gate_grad(a.z, b.z, g.z, gate, da.z, db.z);
    gate_grad(a.w, b.w, g.w, gate, da.w, db.w);
    reinterpret_cast<float4*>(dx)[n * 2 * CL4 + r] = da;
    reinterpret_cast<float4*>(dx)[n * 2 * CL4 + CL4 + r] = db;
  }
}

}  // namespace

#define EW_STREAM ((hipStream_t)stream)

PG_EXPORT int pg_act_fwd(const float* x, float* y, size_t n, int act, void* stream) {
  PG_REQUIRE(x && y, PG_EINVAL, "pg_act_fwd: null pointer");
  if (n == 0) return 0;
  const int vec = aligned16(x) && aligned16(y);
  const int blocks = ew_blocks(vec ? (n + 3) / 4 : n);
  switch (act) {
    case PG_ACT_RELU: hipLaunchKernelGGL(act_fwd_kernel<PG_ACT_RELU>, dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, x, y, n, vec); break;
    case PG_ACT_ELU:  hipLaunchKernelGGL(act_fwd_kernel<PG_ACT_ELU>,  dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, x, y, n, vec); break;
    case PG_ACT_GELU: hipLaunchKernelGGL(act_fwd_kernel<PG_ACT_GELU>, dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, x, y, n, vec); break;
    default: PG_REQUIRE(false, PG_EINVAL, "pg_act_fwd: bad act %d", act);
  }
  PG_LAUNCH_CHECK("pg_act_fwd");
  return 0;
}

PG_EXPORT int pg_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act,
                         void* stream) {
  PG_REQUIRE(x && dy && dx, PG_EINVAL, "pg_act_bwd: null pointer");
  if (n == 0) return 0;
  const int vec = aligned16(x) && aligned16(dy) && aligned16(dx);
  const int blocks = ew_blocks(vec ? (n + 3) / 4 : n);
  switch (act) {
    case PG_ACT_RELU: hipLaunchKernelGGL(act_bwd_kernel<PG_ACT_RELU>, dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, x, dy, dx, n, vec); break;
    case PG_ACT_ELU:  hipLaunchKernelGGL(act_bwd_kernel<PG_ACT_ELU>,  dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, x, dy, dx, n, vec); break;
    case PG_ACT_GELU: hipLaunchKernelGGL(act_bwd_kernel<PG_ACT_GELU>, dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, x, dy, dx, n, vec); break;
    default: PG_REQUIRE(false, PG_EINVAL, "pg_act_bwd: bad act %d", act);
  }
  PG_LAUNCH_CHECK("pg_act_bwd");
  return 0;
}

namespace {
int gated_fwd_impl(const char* who, const float* x, const float* res, float* y, int N, int C, int L,
                   int gate, void* stream) {
  PG_REQUIRE(x && y, PG_EINVAL, "%s: null pointer", who);
  PG_REQUIRE(N > 0 && C > 0 && L > 0, PG_EINVAL, "%s: bad dims", who);
  PG_REQUIRE(gate == PG_GATE_TANH || gate == PG_GATE_IDENTITY, PG_EINVAL, "%s: bad gate", who);
  const size_t CL = (size_t)C * L, total = (size_t)N * CL;
  if ((L % 4) == 0 && aligned16(x) && aligned16(y) && (!res || aligned16(res))) {
    hipLaunchKernelGGL(gated_fwd4_kernel, dim3(ew_blocks(total / 4)), dim3(EW_THREADS), 0, EW_STREAM,
                       x, res, y, CL / 4, total / 4, gate);
  } else {
    PG_REQUIRE(res == nullptr, PG_ESHAPE, "%s: the fused residual needs L %% 4 == 0 and aligned tensors", who);
    hipLaunchKernelGGL(gated_fwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, EW_STREAM, x, y,
                       C, CL, total, gate);
  }
  PG_LAUNCH_CHECK(who);
  return 0;
}
}  // namespace

PG_EXPORT int pg_gated_fwd(const float* x, float* y, int N, int C, int L, int gate, void* stream) {
  return gated_fwd_impl("pg_gated_fwd", x, nullptr, y, N, C, L, gate, stream);
}

PG_EXPORT int pg_gated_fwd_res(const float* x, const float* res, float* y, int N, int C, int L,
                               int gate, void* stream) {
  PG_REQUIRE(res, PG_EINVAL, "pg_gated_fwd_res: null residual");
  return gated_fwd_impl("pg_gated_fwd_res", x, res, y, N, C, L, gate, stream);
}

PG_EXPORT int pg_act_bwd_from_out(const float* y, const float* res, const float* dy, float* dx,
                                  size_t n, int act, void* stream) {
  PG_REQUIRE(y && dy && dx, PG_EINVAL, "pg_act_bwd_from_out: null pointer");
  PG_REQUIRE((n % 4) == 0 && aligned16(y) && aligned16(dy) && aligned16(dx) && (!res || aligned16(res)),
             PG_ESHAPE, "pg_act_bwd_from_out: needs n %% 4 == 0 and 16-byte aligned tensors");
  if (n == 0) return 0;
  const int blocks = ew_blocks(n / 4);
  switch (act) {
    case PG_ACT_RELU: hipLaunchKernelGGL(act_bwd_out_kernel<PG_ACT_RELU>, dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, y, res, dy, dx, n / 4); break;
    case PG_ACT_ELU:  hipLaunchKernelGGL(act_bwd_out_kernel<PG_ACT_ELU>,  dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, y, res, dy, dx, n / 4); break;
    default: PG_REQUIRE(false, PG_EINVAL, "pg_act_bwd_from_out: act %d has no derivative in terms of its output", act);
  }
  PG_LAUNCH_CHECK("pg_act_bwd_from_out");
  return 0;
}

PG_EXPORT int pg_gated_bwd(const float* x, const float* dy, float* dx, int N, int C, int L,
                           int gate, void* stream) {
  PG_REQUIRE(x && dy && dx, PG_EINVAL, "pg_gated_bwd: null pointer");
  PG_REQUIRE(N > 0 && C > 0 && L > 0, PG_EINVAL, "pg_gated_bwd: bad dims");
  PG_REQUIRE(gate == PG_GATE_TANH || gate == PG_GATE_IDENTITY, PG_EINVAL, "pg_gated_bwd: bad gate");
  const size_t CL = (size_t)C * L, total = (size_t)N * CL;
  if ((L % 4) == 0 && aligned16(x) && aligned16(dy) && aligned16(dx))
    hipLaunchKernelGGL(gated_bwd4_kernel, dim3(ew_blocks(total / 4)), dim3(EW_THREADS), 0, EW_STREAM, x,
                       dy, dx, CL / 4, total / 4, gate);
  else
  hipLaunchKernelGGL(gated_bwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, EW_STREAM, x, dy,
                     dx, C, CL, total, gate);
  PG_LAUNCH_CHECK("pg_gated_bwd");
  return 0;
}

PG_EXPORT int pg_add(const float* a, const float* b, float* out, size_t n, void* stream) {
  PG_REQUIRE(a && b && out, PG_EINVAL, "pg_add: null pointer");
  if (n == 0) return 0;
  const int vec = aligned16(a) && aligned16(b) && aligned16(out);
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(vec ? (n + 3) / 4 : n)), dim3(EW_THREADS), 0,
                     EW_STREAM, a, b, out, n, vec);
  PG_LAUNCH_CHECK("pg_add");
  return 0;
}

PG_EXPORT int pg_fill(float* out, float value, size_t n, void* stream) {
  PG_REQUIRE(out, PG_EINVAL, "pg_fill: null pointer");
  if (n == 0) return 0;
  const int vec = aligned16(out);
  hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(vec ? (n + 3) / 4 : n)), dim3(EW_THREADS), 0, EW_STREAM, out, value, n, vec);
  PG_LAUNCH_CHECK("pg_fill");
  return 0;
}

PG_EXPORT int pg_sum_rows(const float* const* rows, const long* batch_strides, int n_rows, float* out, long n_batch, long per,
                          void* stream) {
  PG_REQUIRE(rows && out, PG_EINVAL, "pg_sum_rows: null pointer");
  PG_REQUIRE(n_rows >= 1 && n_rows <= 32, PG_ESHAPE, "pg_sum_rows: 1..32 rows per launch, got %d", n_rows);
  PG_REQUIRE(n_batch >= 0 && per >= 0, PG_EINVAL, "pg_sum_rows: negative size");
  if (n_batch == 0 || per == 0) return 0;
  SumRowsArgs a;
  a.dense = 1;
  for (int k = 0; k < 32; ++k) {
    a.rows[k] = k < n_rows ? rows[k] : nullptr;
    a.bs[k] = (k < n_rows && batch_strides) ? batch_strides[k] : per;
    if (k < n_rows) {
      PG_REQUIRE(a.rows[k], PG_EINVAL, "pg_sum_rows: null row %d", k);
      PG_REQUIRE(a.bs[k] >= per, PG_EINVAL, "pg_sum_rows: batch stride of row %d below the row length", k);
      if (a.bs[k] != per) a.dense = 0;
    }
  }
  a.out = out; a.per = per; a.n = (size_t)n_batch * (size_t)per; a.n_rows = n_rows;
  hipLaunchKernelGGL(sum_rows_kernel, dim3(ew_blocks(a.n)), dim3(EW_THREADS), 0, EW_STREAM, a);
  PG_LAUNCH_CHECK("pg_sum_rows");
  return 0;
}

PG_EXPORT int pg_mul_inplace(float* w, const float* mask, size_t n, void* stream) {
  PG_REQUIRE(w && mask, PG_EINVAL, "pg_mul_inplace: null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(mul_inplace_kernel, dim3(ew_blocks(n)), dim3(EW_THREADS), 0, EW_STREAM, w, mask, n);
  PG_LAUNCH_CHECK("pg_mul_inplace");
  return 0;
}

PG_EXPORT int pg_add_bcast_fwd(const float* x, const float* p, float* y, int N, size_t per,
                               void* stream) {
  PG_REQUIRE(x && p && y, PG_EINVAL, "pg_add_bcast_fwd: null pointer");
  PG_REQUIRE(N > 0 && per > 0, PG_EINVAL, "pg_add_bcast_fwd: bad dims");
  const size_t total = (size_t)N * per;
  hipLaunchKernelGGL(add_bcast_fwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, EW_STREAM, x,
                     p, y, per, total);
  PG_LAUNCH_CHECK("pg_add_bcast_fwd");
  return 0;
}

PG_EXPORT int pg_add_bcast_bwd(const float* dy, float* dp, int N, size_t per, void* stream) {
  PG_REQUIRE(dy && dp, PG_EINVAL, "pg_add_bcast_bwd: null pointer");
  PG_REQUIRE(N > 0 && per > 0, PG_EINVAL, "pg_add_bcast_bwd: bad dims");
  hipLaunchKernelGGL(add_bcast_bwd_kernel, dim3((unsigned)((per + 31) / 32)), dim3(1024), 0, EW_STREAM,
                     dy, dp, N, per);
  PG_LAUNCH_CHECK("pg_add_bcast_bwd");
  return 0;
}

PG_EXPORT int pg_image_positional_encoding(float* out, int N, int H, int W, void* stream) {
  PG_REQUIRE(out, PG_EINVAL, "pg_image_positional_encoding: null pointer");
  PG_REQUIRE(N > 0 && H > 0 && W > 0, PG_EINVAL, "pg_image_positional_encoding: bad dims");
  const size_t total = (size_t)N * 2 * H * W;
  hipLaunchKernelGGL(posenc_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, EW_STREAM, out, N, H, W);
  PG_LAUNCH_CHECK("pg_image_positional_encoding");
  return 0;
}

PG_EXPORT int pg_bce_logits_fwd(const float* z, const float* x, float* loss, int N, size_t per,
                                void* stream) {
  PG_REQUIRE(z && x && loss, PG_EINVAL, "pg_bce_logits_fwd: null pointer");
  PG_REQUIRE(N > 0 && per > 0, PG_EINVAL, "pg_bce_logits_fwd: bad dims");
  const size_t total = (size_t)N * per;
  int blocks = ew_blocks(total);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(blocks), dim3(EW_THREADS), 0, EW_STREAM, z, x, loss, total,
                     1.f / (float)N);
  PG_LAUNCH_CHECK("pg_bce_logits_fwd");
  return 0;
}

PG_EXPORT int pg_bce_logits_bwd(const float* z, const float* x, const float* gscale, float* dz,
                                int N, size_t per, void* stream) {
  PG_REQUIRE(z && x && gscale && dz, PG_EINVAL, "pg_bce_logits_bwd: null pointer");
  PG_REQUIRE(N > 0 && per > 0, PG_EINVAL, "pg_bce_logits_bwd: bad dims");
  const size_t total = (size_t)N * per;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, EW_STREAM, z, x,
                     gscale, dz, total, 1.f / (float)N);
  PG_LAUNCH_CHECK("pg_bce_logits_bwd");
  return 0;
}

// ---- strided row copy ---------------------------------------------------------------------------
// dst[r * dst_stride + i] (+)= src[r * src_stride + i], r < rows, i < row_len. One kernel for the
// channel concatenation in front of PixelSNAIL's merged [q | k | v] projection (reference
// nn/attention.py:139-143: torch.cat((x, extra_x), dim=1); a row = the channels of one image) and for
// assembling / splitting that projection's merged weight (rows of different widths).
namespace {
__global__ void copy_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long rows,
                                 long row_len, long src_stride, long dst_stride, int accumulate, int vec) {
  const long per = vec ? row_len / 4 : row_len;
  const long total = rows * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / per, c = i - r * per;
    if (vec) {
      const float4 v = *reinterpret_cast<const float4*>(src + r * src_stride + 4 * c);
      float4* d = reinterpret_cast<float4*>(dst + r * dst_stride + 4 * c);
      if (accumulate) {
        const float4 o = *d;
        *d = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
      } else {
        *d = v;
      }
    } else {
      float* d = dst + r * dst_stride + c;
      *d = accumulate ? *d + src[r * src_stride + c] : src[r * src_stride + c];
    }
  }
}
}  // namespace

PG_EXPORT int pg_copy_rows(const float* src, float* dst, long rows, long row_len, long src_stride,
                           long dst_stride, int accumulate, void* stream) {
  PG_REQUIRE(src && dst, PG_EINVAL, "pg_copy_rows: null pointer");
  PG_REQUIRE(rows >= 0 && row_len >= 0 && src_stride >= row_len && dst_stride >= row_len, PG_EINVAL,
             "pg_copy_rows: bad extents");
  if (rows == 0 || row_len == 0) return 0;
  const int vec = (row_len % 4 == 0 && src_stride % 4 == 0 && dst_stride % 4 == 0 &&
                   (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) ? 1 : 0;
  const long total = rows * (vec ? row_len / 4 : row_len);
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst,
                     rows, row_len, src_stride, dst_stride, accumulate, vec);
  PG_LAUNCH_CHECK("pg_copy_rows");
  return 0;
}

// ---- concat_elu (PixelCNN++: Salimans et al. 2017, section 2.3 "concatenated ELU") -------------------
// y[:, :C] = elu(x), y[:, C:] = elu(-x); dx = dy1 * elu'(x) - dy2 * elu'(-x). x (N, C, L), y (N, 2C, L).
namespace {
__global__ void concat_elu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long total, long CL) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / CL, r = i - n * CL;
    const float v = x[i];
    y[n * 2 * CL + r] = pg_apply_act(v, PG_ACT_ELU);
    y[n * 2 * CL + CL + r] = pg_apply_act(-v, PG_ACT_ELU);
  }
}
__global__ void concat_elu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                      float* __restrict__ dx, long total, long CL) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / CL, r = i - n * CL;
    const float v = x[i];
    dx[i] = dy[n * 2 * CL + r] * pg_act_grad(v, PG_ACT_ELU) - dy[n * 2 * CL + CL + r] * pg_act_grad(-v, PG_ACT_ELU);
  }
}
}  // namespace

PG_EXPORT int pg_concat_elu_fwd(const float* x, float* y, int N, long CL, void* stream) {
  PG_REQUIRE(x && y && N > 0 && CL > 0, PG_EINVAL, "pg_concat_elu_fwd: bad arguments");
  const long total = (long)N * CL;
  hipLaunchKernelGGL(concat_elu_fwd_kernel, dim3(ew_blocks((size_t)total)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     x, y, total, CL);
  PG_LAUNCH_CHECK("pg_concat_elu_fwd");
  return 0;
}

PG_EXPORT int pg_concat_elu_bwd(const float* x, const float* dy, float* dx, int N, long CL, void* stream) {
  PG_REQUIRE(x && dy && dx && N > 0 && CL > 0, PG_EINVAL, "pg_concat_elu_bwd: bad arguments");
  const long total = (long)N * CL;
  hipLaunchKernelGGL(concat_elu_bwd_kernel, dim3(ew_blocks((size_t)total)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     x, dy, dx, total, CL);
  PG_LAUNCH_CHECK("pg_concat_elu_bwd");
  return 0;
}

