// vae_ops.hip — the non-convolutional pieces of the VAE encoder/decoder stacks (fp32).
//
// Reference call sites: AvgPool2d(2,2) / Upsample(nearest, x2) in models/vae/vd_vae.py:224,256;
// unit_gaussian_kl_div / gaussian_kl_div / sample_from_gaussian in models/vae/vaes.py:17-33 as
// used by vae.py:91-93 and vd_vae.py:158-189; the ELBO `recon + kl` mean of vae.py:149-159.
// The Gaussian heads are fused: one kernel reads the [mean | log_std] channel halves of the
// producing conv's output IN PLACE (no split copies), writes z = mu + exp(s) * eps and reduces
// the per-sample KL sum; the backward kernel writes the gradient of the whole conv output.
#include "common.h"

namespace {

constexpr int VT = 256;

inline int vblocks(size_t n) {
  size_t b = (n + VT - 1) / VT;
  if (b > 4096) b = 4096;
  return (int)(b < 1 ? 1 : b);
}

// ---- 2x2 average pool / nearest x2 upsample (H, W even) -------------------------------------
__global__ void avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total,
                                    int OH, int OW) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % OW);
    const size_t t = i / OW;
    const int r = (int)(t % OH);
    const size_t plane = t / OH;
    const float* p = x + (plane * 2 * OH + 2 * r) * (size_t)(2 * OW) + 2 * c;
    y[i] = 0.25f * ((p[0] + p[1]) + (p[2 * OW] + p[2 * OW + 1]));
  }
}

// dx[2r+i, 2c+j] = 0.25 * dy[r, c]   (also the forward of "nearest upsample" with scale 1)
__global__ void expand2_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t total,
                               int OH, int OW, float scale, const float* __restrict__ res) {  // total = elements of dst
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % (2 * OW));
    const size_t t = i / (2 * OW);
    const int r = (int)(t % (2 * OH));
    const size_t plane = t / (2 * OH);
    const float v = scale * src[(plane * OH + (r >> 1)) * (size_t)OW + (c >> 1)];
    dst[i] = res ? v + res[i] : v;
  }
}

// dx[r, c] = sum of the 2x2 block of dy  (backward of nearest upsample)
__global__ void sum2x2_kernel(const float* __restrict__ dy, float* __restrict__ dx, size_t total,
                              int OH, int OW) {  // total = elements of dx (OH x OW planes)
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % OW);
    const size_t t = i / OW;
    const int r = (int)(t % OH);
    const size_t plane = t / OH;
    const float* p = dy + (plane * 2 * OH + 2 * r) * (size_t)(2 * OW) + 2 * c;
    dx[i] = (p[0] + p[1]) + (p[2 * OW] + p[2 * OW + 1]);
  }
}

// ---- 2x2 phase split / merge (space-to-depth): x (planes, 2H, 2W) <-> xs (4, planes, H, W) with
// xs[2*pr + pc][plane][r][c] = x[plane][2r + pr][2c + pc]. Stride-2 4x4 convolutions and their
// transposes (vaes.py:153-160, 228-235) are evaluated as four stride-1 2x2 tap convolutions on
// the phases.
__global__ void phase_split_kernel(float* x, float* xs, size_t total,
                                   int H, int W, size_t phase_elems, int merge) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;  // total = planes * 2H * 2W
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c2 = (int)(i % (2 * W));
    const size_t t = i / (2 * W);
    const int r2 = (int)(t % (2 * H));
    const size_t plane = t / (2 * H);
    const size_t j = (size_t)(2 * (r2 & 1) + (c2 & 1)) * phase_elems +
                     (plane * H + (r2 >> 1)) * (size_t)W + (c2 >> 1);
    if (merge) x[i] = xs[j];
    else xs[j] = x[i];
  }
}

struct Phase4 { float* p[4]; };

__global__ void phase_merge4_kernel(float* x, const Phase4 ph, size_t total, int H, int W, int merge) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;  // total = planes * 2H * 2W
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c2 = (int)(i % (2 * W));
    const size_t t = i / (2 * W);
    const int r2 = (int)(t % (2 * H));
    const size_t plane = t / (2 * H);
    float* q = ph.p[2 * (r2 & 1) + (c2 & 1)] + (plane * H + (r2 >> 1)) * (size_t)W + (c2 >> 1);
    if (merge) x[i] = *q;
    else *q = x[i];
  }
}

// one thread per element of the (4, Co, Ci, 2, 2) phase-kernel tensor; `src` = its element of the 4x4 weight (a bijection)
struct PhaseW { const float* g[4]; };
__device__ __forceinline__ size_t phase_w_src(int idx, int A, int B, int transposed, int& k, int& rest) {
  const int Co = transposed ? B : A, Ci = transposed ? A : B;
  const int j = idx & 1, i = (idx >> 1) & 1;
  int t = idx >> 2;
  const int c = t % Ci; t /= Ci;
  const int o = t % Co;
  k = t / Co;
  rest = ((o * Ci + c) * 2 + i) * 2 + j;
  const int pr = k >> 1, pc = k & 1;
  if (transposed) return (((size_t)c * B + o) * 4 + (2 * (1 - i) + 1 - pr)) * 4 + (2 * (1 - j) + 1 - pc);
  return (((size_t)o * B + c) * 4 + (2 * i + 1 - pr)) * 4 + (2 * j + 1 - pc);
}

__global__ void phase_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int total, int A, int B, int transposed) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int k, rest;
    out[idx] = w[phase_w_src(idx, A, B, transposed, k, rest)];
  }
}

__global__ void phase_weights_bwd_kernel(const PhaseW g, float* __restrict__ dw, int total, int A, int B, int transposed,
                                         int accumulate) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int k, rest;
    const size_t src = phase_w_src(idx, A, B, transposed, k, rest);
    const float v = g.g[k] ? g.g[k][rest] : 0.f;
    dw[src] = accumulate ? dw[src] + v : v;
  }
}

// ---- Gaussian heads --------------------------------------------------------------------------
// q: (N, >=2C, L) with [mean | log_std] in channels [0,C) and [C,2C); p: same layout or NULL.
//   p == NULL : KL(q || N(0,1)) = -0.5 (1 + 2 s - e^{2s} - mu^2)                 vaes.py:17-19
//   p != NULL : KL(q || p) = -0.5 + (sp - sq) + (e^{2 sq} + (mq - mp)^2) / (2 e^{2 sp})  vaes.py:23-27
// z = mu_q + exp(s_q) * eps  (vaes.py:31-33).   kl[n] += sum over (c, l).
// mode 2 (prior sampling, vd_vae.py:166-168): z = mu_p + exp(s_p) * eps, no KL (q unused).
struct GaussArgs {
  const float* q; const float* p; const float* eps; float* z; float* kl;
  const float* dz; const float* dkl; float* dq; float* dp;
  int N, C, L, mode;
  long q_bs, p_bs;  // batch strides in floats
};

__global__ void __launch_bounds__(VT) gauss_fwd_kernel(const GaussArgs a) {
  const int n = blockIdx.y;
  const size_t CL = (size_t)a.C * a.L;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * VT + threadIdx.x; i < CL; i += (size_t)gridDim.x * VT) {
    const float e = a.eps[(size_t)n * CL + i];
    if (a.mode == 2) {
      const float mp = a.p[(size_t)n * a.p_bs + i], sp = a.p[(size_t)n * a.p_bs + CL + i];
      a.z[(size_t)n * CL + i] = fmaf(expf(sp), e, mp);
      continue;
    }
    const float mq = a.q[(size_t)n * a.q_bs + i], sq = a.q[(size_t)n * a.q_bs + CL + i];
    a.z[(size_t)n * CL + i] = fmaf(expf(sq), e, mq);
    if (a.mode == 0) {
      acc += -0.5f * (1.f + 2.f * sq - expf(2.f * sq) - mq * mq);
    } else {
      const float mp = a.p[(size_t)n * a.p_bs + i], sp = a.p[(size_t)n * a.p_bs + CL + i];
      const float d = mq - mp;
      acc += -0.5f + (sp - sq) + (expf(2.f * sq) + d * d) / (2.f * expf(2.f * sp));
    }
  }
  if (a.mode == 2) return;
  acc = pg_wave_sum(acc);
  __shared__ float part[VT / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < VT / 64; ++w) t += part[w];
    atomicAdd(&a.kl[n], t);
  }
}

// dq / dp cover the [mean | log_std] channels; dp's remaining channels (if any) are zeroed by the caller.
__global__ void __launch_bounds__(VT) gauss_bwd_kernel(const GaussArgs a) {
  const int n = blockIdx.y;
  const size_t CL = (size_t)a.C * a.L;
  const float gk = (a.mode != 2 && a.dkl) ? a.dkl[n] : 0.f;
  for (size_t i = (size_t)blockIdx.x * VT + threadIdx.x; i < CL; i += (size_t)gridDim.x * VT) {
    const float e = a.eps[(size_t)n * CL + i];
    const float gz = a.dz ? a.dz[(size_t)n * CL + i] : 0.f;
    if (a.mode == 2) {
      const float sp = a.p[(size_t)n * a.p_bs + CL + i];
      a.dp[(size_t)n * a.p_bs + i] = gz;
      a.dp[(size_t)n * a.p_bs + CL + i] = gz * e * expf(sp);
      continue;
    }
    const float mq = a.q[(size_t)n * a.q_bs + i], sq = a.q[(size_t)n * a.q_bs + CL + i];
    float dmq = gz, dsq = gz * e * expf(sq);
    if (a.mode == 0) {
      dmq += gk * mq;
      dsq += gk * (expf(2.f * sq) - 1.f);
    } else {
      const float mp = a.p[(size_t)n * a.p_bs + i], sp = a.p[(size_t)n * a.p_bs + CL + i];
      const float iv = expf(-2.f * sp), vq = expf(2.f * sq), d = mq - mp;
      dmq += gk * d * iv;
      dsq += gk * (vq * iv - 1.f);
      a.dp[(size_t)n * a.p_bs + i] = -gk * d * iv;
      a.dp[(size_t)n * a.p_bs + CL + i] = gk * (1.f - (vq + d * d) * iv);
    }
    a.dq[(size_t)n * a.q_bs + i] = dmq;
    a.dq[(size_t)n * a.q_bs + CL + i] = dsq;
  }
}

__global__ void vec_mean_accum_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += v[i];
  s = pg_wave_sum(s);
  if (threadIdx.x == 0) out[0] += s / (float)n;
}

__global__ void fill_scaled_kernel(const float* __restrict__ g, float scale, float* __restrict__ out, int n) {
  for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x) out[i] = g[0] * scale;
}

int check_gauss(const char* who, int N, int C, int L) {
  PG_REQUIRE(N > 0 && C > 0 && L > 0, PG_EINVAL, "%s: non-positive dimension", who);
  PG_REQUIRE(N <= 65535, PG_ESHAPE, "%s: batch exceeds the grid limit", who);
  return 0;
}

}  // namespace

#define VST ((hipStream_t)stream)

PG_EXPORT int pg_avgpool2_fwd(const float* x, float* y, int planes, int OH, int OW, void* stream) {
  PG_REQUIRE(x && y && planes > 0 && OH > 0 && OW > 0, PG_EINVAL, "pg_avgpool2_fwd: bad arguments");
  const size_t total = (size_t)planes * OH * OW;
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, x, y, total, OH, OW);
  PG_LAUNCH_CHECK("pg_avgpool2_fwd");
  return 0;
}

PG_EXPORT int pg_avgpool2_bwd(const float* dy, float* dx, int planes, int OH, int OW, void* stream) {
  PG_REQUIRE(dy && dx && planes > 0 && OH > 0 && OW > 0, PG_EINVAL, "pg_avgpool2_bwd: bad arguments");
  const size_t total = (size_t)planes * OH * OW * 4;
  hipLaunchKernelGGL(expand2_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, dy, dx, total, OH, OW, 0.25f, (const float*)nullptr);
  PG_LAUNCH_CHECK("pg_avgpool2_bwd");
  return 0;
}

PG_EXPORT int pg_avgpool2_bwd_res(const float* dy, const float* res, float* dx, int planes, int OH, int OW, void* stream) {
  PG_REQUIRE(dy && res && dx && planes > 0 && OH > 0 && OW > 0, PG_EINVAL, "pg_avgpool2_bwd_res: bad arguments");
  const size_t total = (size_t)planes * OH * OW * 4;
  hipLaunchKernelGGL(expand2_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, dy, dx, total, OH, OW, 0.25f, res);
  PG_LAUNCH_CHECK("pg_avgpool2_bwd_res");
  return 0;
}

PG_EXPORT int pg_upsample2_fwd(const float* x, float* y, int planes, int IH, int IW, void* stream) {
  PG_REQUIRE(x && y && planes > 0 && IH > 0 && IW > 0, PG_EINVAL, "pg_upsample2_fwd: bad arguments");
  const size_t total = (size_t)planes * IH * IW * 4;
  hipLaunchKernelGGL(expand2_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, x, y, total, IH, IW, 1.0f, (const float*)nullptr);
  PG_LAUNCH_CHECK("pg_upsample2_fwd");
  return 0;
}

PG_EXPORT int pg_upsample2_bwd(const float* dy, float* dx, int planes, int IH, int IW, void* stream) {
  PG_REQUIRE(dy && dx && planes > 0 && IH > 0 && IW > 0, PG_EINVAL, "pg_upsample2_bwd: bad arguments");
  const size_t total = (size_t)planes * IH * IW;
  hipLaunchKernelGGL(sum2x2_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, dy, dx, total, IH, IW);
  PG_LAUNCH_CHECK("pg_upsample2_bwd");
  return 0;
}

/* merge == 0: xs <- split(x); merge == 1: x <- merge(xs).  x: (planes, 2H, 2W), xs: (4, planes, H, W) */
PG_EXPORT int pg_phase_split2(float* x, float* xs, int planes, int H, int W, int merge, void* stream) {
  PG_REQUIRE(x && xs && planes > 0 && H > 0 && W > 0, PG_EINVAL, "pg_phase_split2: bad arguments");
  const size_t total = (size_t)planes * 4 * H * W;
  hipLaunchKernelGGL(phase_split_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, x, xs, total, H, W,
                     (size_t)planes * H * W, merge);
  PG_LAUNCH_CHECK("pg_phase_split2");
  return 0;
}

PG_EXPORT int pg_phase_merge4(float* x, float* const* p, int planes, int H, int W, int merge, void* stream) {
  PG_REQUIRE(x && p && p[0] && p[1] && p[2] && p[3] && planes > 0 && H > 0 && W > 0, PG_EINVAL, "pg_phase_merge4: bad arguments");
  Phase4 ph;
  for (int k = 0; k < 4; ++k) ph.p[k] = p[k];
  const size_t total = (size_t)planes * 4 * H * W;
  hipLaunchKernelGGL(phase_merge4_kernel, dim3(vblocks(total)), dim3(VT), 0, VST, x, ph, total, H, W, merge);
  PG_LAUNCH_CHECK("pg_phase_merge4");
  return 0;
}

PG_EXPORT int pg_phase_weights(const float* w, float* out, int A, int B, int transposed, void* stream) {
  PG_REQUIRE(w && out && A > 0 && B > 0, PG_EINVAL, "pg_phase_weights: bad arguments");
  const long total = 16L * A * B;
  PG_REQUIRE(total < (1L << 30), PG_ESHAPE, "pg_phase_weights: weight too large");
  hipLaunchKernelGGL(phase_weights_kernel, dim3(vblocks((size_t)total)), dim3(VT), 0, VST, w, out, (int)total, A, B, transposed);
  PG_LAUNCH_CHECK("pg_phase_weights");
  return 0;
}

PG_EXPORT int pg_phase_weights_bwd(const float* const* g, float* dw, int A, int B, int transposed, int accumulate, void* stream) {
  PG_REQUIRE(g && dw && A > 0 && B > 0, PG_EINVAL, "pg_phase_weights_bwd: bad arguments");
  const long total = 16L * A * B;
  PG_REQUIRE(total < (1L << 30), PG_ESHAPE, "pg_phase_weights_bwd: weight too large");
  PhaseW pw;
  for (int k = 0; k < 4; ++k) pw.g[k] = g[k];
  hipLaunchKernelGGL(phase_weights_bwd_kernel, dim3(vblocks((size_t)total)), dim3(VT), 0, VST, pw, dw, (int)total, A, B, transposed,
                     accumulate);
  PG_LAUNCH_CHECK("pg_phase_weights_bwd");
  return 0;
}

PG_EXPORT int pg_gauss_head_fwd(const float* q, const float* p, const float* eps, float* z, float* kl,
                                int N, int C, int L, long q_bs, long p_bs, int mode, void* stream) {
  int rc = check_gauss("pg_gauss_head_fwd", N, C, L);
  if (rc) return rc;
  PG_REQUIRE(eps && z && mode >= 0 && mode <= 2, PG_EINVAL, "pg_gauss_head_fwd: bad arguments");
  PG_REQUIRE((mode == 2 || (q && kl)) && (mode == 0 || p), PG_EINVAL, "pg_gauss_head_fwd: null pointer");
  GaussArgs a = {};
  a.q = q; a.p = p; a.eps = eps; a.z = z; a.kl = kl; a.N = N; a.C = C; a.L = L; a.mode = mode;
  a.q_bs = q_bs; a.p_bs = p_bs;
  const size_t CL = (size_t)C * L;
  int bx = (int)((CL + VT - 1) / VT);
  if (bx > 64) bx = 64;
  // bit-reproducible mode (pg_attn_fused_bwd(0) = ops.set_deterministic): ONE block per sample, so that kl[n] is a fixed-order
  // sum instead of an arrival-order sum of up to 64 block partials (last-bit differences run to run, found by the twin-process
  // self-check of round 6; the value feeds no gradient — d loss / d kl is a constant — but it is a module output)
  if (pg_attn_fused_bwd(-1) == 0) bx = 1;
  hipLaunchKernelGGL(gauss_fwd_kernel, dim3(bx, N), dim3(VT), 0, VST, a);
  PG_LAUNCH_CHECK("pg_gauss_head_fwd");
  return 0;
}

PG_EXPORT int pg_gauss_head_bwd(const float* q, const float* p, const float* eps, const float* dz,
                                const float* dkl, float* dq, float* dp, int N, int C, int L, long q_bs,
                                long p_bs, int mode, void* stream) {
  int rc = check_gauss("pg_gauss_head_bwd", N, C, L);
  if (rc) return rc;
  PG_REQUIRE(eps && mode >= 0 && mode <= 2, PG_EINVAL, "pg_gauss_head_bwd: bad arguments");
  PG_REQUIRE((mode == 2 || (q && dq)) && (mode == 0 || (p && dp)), PG_EINVAL, "pg_gauss_head_bwd: null pointer");
  GaussArgs a = {};
  a.q = q; a.p = p; a.eps = eps; a.dz = dz; a.dkl = dkl; a.dq = dq; a.dp = dp;
  a.N = N; a.C = C; a.L = L; a.mode = mode; a.q_bs = q_bs; a.p_bs = p_bs;
  const size_t CL = (size_t)C * L;
  int bx = (int)((CL + VT - 1) / VT);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(gauss_bwd_kernel, dim3(bx, N), dim3(VT), 0, VST, a);
  PG_LAUNCH_CHECK("pg_gauss_head_bwd");
  return 0;
}

/* out[0] += mean(v[0..n)) */
PG_EXPORT int pg_vec_mean_accum(const float* v, int n, float* out, void* stream) {
  PG_REQUIRE(v && out && n > 0, PG_EINVAL, "pg_vec_mean_accum: bad arguments");
  hipLaunchKernelGGL(vec_mean_accum_kernel, dim3(1), dim3(64), 0, VST, v, n, out);
  PG_LAUNCH_CHECK("pg_vec_mean_accum");
  return 0;
}

/* out[i] = g[0] * scale, i < n  (the backward of a mean) */
PG_EXPORT int pg_fill_scaled(const float* g, float scale, float* out, int n, void* stream) {
  PG_REQUIRE(g && out && n > 0, PG_EINVAL, "pg_fill_scaled: bad arguments");
  hipLaunchKernelGGL(fill_scaled_kernel, dim3((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), dim3(256), 0, VST, g, scale, out, n);
  PG_LAUNCH_CHECK("pg_fill_scaled");
  return 0;
}
