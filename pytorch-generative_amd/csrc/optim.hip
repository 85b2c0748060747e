// optim.hip — the optimiser half of the timed step (reference trainer.py:183-191):
// clip_grad_norm_ (global L2 norm over every gradient) followed by torch.optim.Adam, over ONE
// flat fp32 parameter / gradient / moment buffer, driven entirely by a small device-side
// state block so the whole step replays from a hipGraph with no host round trip.
//
// state[0]=step  [1]=lr  [2]=sumsq scratch  [3]=grad norm (out)  [4]=clip coef (out)
// state[5]=lr multiplier applied after each step (MultiplicativeLR, image_gpt.py:156)
// state[6]=max_norm (1e50 when unset, trainer.py:183)  [7]=grad pre-scale (1/world)
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;

__global__ void __launch_bounds__(OPT_THREADS)
sumsq_kernel(const float* __restrict__ g, size_t n, float* __restrict__ state) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  const size_t n4 = n >> 2;
  const bool vec = (((uintptr_t)g) & 15) == 0;
  if (vec) {
    for (; i < n4; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(g)[i];
      s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s);
    }
    for (size_t t = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
      s = fmaf(g[t], g[t], s);
  } else {
    for (; i < n; i += stride) s = fmaf(g[i], g[i], s);
  }
  s = pg_wave_sum(s);
  __shared__ float part[OPT_THREADS / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < OPT_THREADS / 64; ++w) t += part[w];
    atomicAdd(&state[2], t);
  }
}

// one thread: finish the norm, derive the clip coefficient, advance step.
// clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1.
__global__ void adam_prepare_kernel(float* __restrict__ state) {
  const float pre = state[7];
  const float norm = sqrtf(state[2]) * pre;
  state[3] = norm;
  float coef = state[6] / (norm + 1e-6f);
  coef = coef > 1.f ? 1.f : coef;
  state[4] = coef * pre;
  state[0] = state[0] + 1.f;
  state[2] = 0.f;
}

__global__ void __launch_bounds__(OPT_THREADS)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, size_t n, const float* __restrict__ state, float beta1,
            float beta2, float eps) {
  const float step = state[0];
  const float lr = state[1];
  const float gscale = state[4];
  // torch.optim.Adam (non-amsgrad, no weight decay):
  //   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2
  //   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
  const float bc1 = 1.f - powf(beta1, step);
  const float bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i] * gscale;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

__global__ void lr_decay_kernel(float* __restrict__ state) { state[1] *= state[5]; }

}  // namespace

PG_EXPORT int pg_sumsq_accum(const float* g, size_t n, float* state, void* stream) {
  PG_REQUIRE(g && state, PG_EINVAL, "pg_sumsq_accum: null pointer");
  if (n == 0) return 0;
  size_t blocks = ((n + 3) / 4 + OPT_THREADS - 1) / OPT_THREADS;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream,
                     g, n, state);
  PG_LAUNCH_CHECK("pg_sumsq_accum");
  return 0;
}

PG_EXPORT int pg_adam_prepare(float* state, void* stream) {
  PG_REQUIRE(state, PG_EINVAL, "pg_adam_prepare: null pointer");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
  PG_LAUNCH_CHECK("pg_adam_prepare");
  return 0;
}

PG_EXPORT int pg_adam_step(float* p, const float* g, float* m, float* v, size_t n,
                           const float* state, float beta1, float beta2, float eps,
                           void* stream) {
  PG_REQUIRE(p && g && m && v && state, PG_EINVAL, "pg_adam_step: null pointer");
  if (n == 0) return 0;
  size_t blocks = (n + OPT_THREADS - 1) / OPT_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(OPT_THREADS), 0, (hipStream_t)stream,
                     p, g, m, v, n, state, beta1, beta2, eps);
  hipLaunchKernelGGL(lr_decay_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                     const_cast<float*>(state));
  PG_LAUNCH_CHECK("pg_adam_step");
  return 0;
}
