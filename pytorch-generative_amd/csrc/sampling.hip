// sampling.hip — the two extra kernels of incremental autoregressive sampling.
//
// Reference: AutoregressiveModel.sample (models/base.py:97-120) runs ONE FULL forward per pixel
// (H*W forwards per batch of samples; nn/attention.py:198-202 carries the author's TODO about it).
// The models are causal, so the activations of pixel p are final once pixels < p are: the sampler in
// models/image_gpt.py evaluates only position p per step, on a (channels, batch) matrix — which is
// exactly the (C, L) plane layout of the block kernels with the batch as the pixel axis, so
// gpt_block.hip's head / tail kernels are reused unchanged — and keeps per-layer K / V caches.
//   pg_sample_embed:  one output pixel of the masked input convolution on `canvas + pos`
//   pg_attn_decode:   one query per (n, head) against the cached keys / values of positions <= p
#include "common.h"

namespace {

// out[co * ld + n] = b[co] + sum_{ci,u,v} w[co][ci][u][v] * xin(n, ci, r + u - ph, c + v - pw),
// xin = canvas + pos inside the image, 0 outside (zero padding applies to x + pos, image_gpt.py:105).
// The weight is already masked in place (nn/convolution.py:42), so every tap is read.
__global__ void sample_embed_kernel(const float* __restrict__ canvas, const float* __restrict__ pos,
                                    const float* __restrict__ w, const float* __restrict__ b,
                                    float* __restrict__ out, int N, int Cin, int H, int W, int Cout,
                                    int KH, int KW, int r, int c, int ld, const int* __restrict__ pos_dev) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int co = blockIdx.y;
  if (n >= N) return;
  if (pos_dev) {  // raster position kept on the device: the step is replayed from one hipGraph
    const int p = min(max(*pos_dev, 0), H * W - 1);
    r = p / W;
    c = p - r * W;
  }
  float acc = b ? b[co] : 0.f;
  for (int ci = 0; ci < Cin; ++ci) {
    for (int u = 0; u < KH; ++u) {
      const int rr = r + u - KH / 2;
      if (rr < 0 || rr >= H) continue;
      for (int v = 0; v < KW; ++v) {
        const int cc = c + v - KW / 2;
        if (cc < 0 || cc >= W) continue;
        const float x = canvas[((size_t)n * Cin + ci) * H * W + (size_t)rr * W + cc] +
                        (pos ? pos[(size_t)ci * H * W + (size_t)rr * W + cc] : 0.f);
        acc = fmaf(w[(((size_t)co * Cin + ci) * KH + u) * KW + v], x, acc);
      }
    }
  }
  out[(size_t)co * ld + n] = acc;
}

// One wave per (head, n). qkv: rows [q (E) | k (E) | v (V)] x ld columns (column n = sample n).
// Appends this position's k / v to the caches (N, E, L) / (N, V, L) at column p and writes
// o[(h*dv + c) * ld + n] = softmax_{j <= p - strict}(q . k_j / sqrt(dk)) v_j  (0 if no key is allowed).
template <int DK, int DV>
__global__ void __launch_bounds__(64) attn_decode_kernel(const float* __restrict__ qkv, float* __restrict__ kc,
                                                         float* __restrict__ vc, float* __restrict__ o,
                                                         int N, int heads, int L, int p, int strict,
                                                         int ld, int dk, int dv, float scale2,
                                                         const int* __restrict__ pos_dev) {
  const int h = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  if (pos_dev) p = min(max(*pos_dev, 0), L - 1);
  const int E = heads * dk, V = heads * dv;
  float q[DK], knew[DK], vnew[DV], acc[DV];
#pragma unroll
  for (int d = 0; d < DK; ++d) {
    q[d] = d < dk ? qkv[(size_t)(h * dk + d) * ld + n] * scale2 : 0.f;
    knew[d] = d < dk ? qkv[(size_t)(E + h * dk + d) * ld + n] : 0.f;
  }
#pragma unroll
  for (int cidx = 0; cidx < DV; ++cidx) {
    vnew[cidx] = cidx < dv ? qkv[(size_t)(2 * E + h * dv + cidx) * ld + n] : 0.f;
    acc[cidx] = 0.f;
  }
  float* kp = kc + ((size_t)n * E + h * dk) * L;
  float* vp = vc + ((size_t)n * V + h * dv) * L;
  if (lane == 0) {
#pragma unroll
    for (int d = 0; d < DK; ++d)
      if (d < dk) kp[(size_t)d * L + p] = knew[d];
#pragma unroll
    for (int cidx = 0; cidx < DV; ++cidx)
      if (cidx < dv) vp[(size_t)cidx * L + p] = vnew[cidx];
  }
  // cached positions j < p (always allowed), lane-local online softmax in the log2 domain
  float m = -1.0e30f, l = 0.f;
  for (int j = lane; j < p; j += 64) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DK; ++d)
      if (d < dk) s = fmaf(q[d], kp[(size_t)d * L + j], s);
    const float mn = fmaxf(m, s);
    const float a = __builtin_amdgcn_exp2f(m - mn), pr = __builtin_amdgcn_exp2f(s - mn);
    l = l * a + pr;
#pragma unroll
    for (int cidx = 0; cidx < DV; ++cidx)
      if (cidx < dv) acc[cidx] = fmaf(pr, vp[(size_t)cidx * L + j], acc[cidx] * a);
    m = mn;
  }
  // the position itself (from registers: its cache entry was written by this very wave)
  if (!strict && lane == 0) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DK; ++d) s = fmaf(q[d], knew[d], s);
    const float mn = fmaxf(m, s);
    const float a = __builtin_amdgcn_exp2f(m - mn), pr = __builtin_amdgcn_exp2f(s - mn);
    l = l * a + pr;
#pragma unroll
    for (int cidx = 0; cidx < DV; ++cidx) acc[cidx] = fmaf(pr, vnew[cidx], acc[cidx] * a);
    m = mn;
  }
  // merge the 64 lanes
  float mx = m;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  const float f = __builtin_amdgcn_exp2f(m - mx);  // lanes without keys: exp2(-1e30 - mx) = 0 or (all empty) 1 * l=0
  l = pg_wave_sum(l * f);
  const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int cidx = 0; cidx < DV; ++cidx) {
    const float t = pg_wave_sum(acc[cidx] * f);
    if (lane == 0 && cidx < dv) o[(size_t)(h * dv + cidx) * ld + n] = t * inv;
  }
}

}  // namespace

PG_EXPORT int pg_sample_embed(const float* canvas, const float* pos, const float* w, const float* b,
                              float* out, int N, int Cin, int H, int W, int Cout, int KH, int KW, int r,
                              int c, int ld, const int* pos_dev, void* stream) {
  PG_REQUIRE(canvas && w && out, PG_EINVAL, "pg_sample_embed: null pointer");
  PG_REQUIRE(N > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && KH > 0 && KW > 0 && ld >= N, PG_EINVAL,
             "pg_sample_embed: bad dims");
  PG_REQUIRE(r >= 0 && r < H && c >= 0 && c < W, PG_EINVAL, "pg_sample_embed: pixel outside the image");
  hipLaunchKernelGGL(sample_embed_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)Cout), dim3(64), 0,
                     (hipStream_t)stream, canvas, pos, w, b, out, N, Cin, H, W, Cout, KH, KW, r, c, ld, pos_dev);
  PG_LAUNCH_CHECK("pg_sample_embed");
  return 0;
}

PG_EXPORT int pg_attn_decode(const float* qkv, float* k_cache, float* v_cache, float* o, int N, int heads,
                             int L, int p, int dk, int dv, int ld, int strict, const int* pos_dev,
                             void* stream) {
  PG_REQUIRE(qkv && k_cache && v_cache && o, PG_EINVAL, "pg_attn_decode: null pointer");
  PG_REQUIRE(N > 0 && heads > 0 && L > 0 && dk > 0 && dv > 0 && ld >= N, PG_EINVAL, "pg_attn_decode: bad dims");
  PG_REQUIRE(p >= 0 && p < L, PG_EINVAL, "pg_attn_decode: position outside the sequence");
  PG_REQUIRE(dk <= 32 && dv <= 32, PG_ESHAPE, "pg_attn_decode: head dims (%d,%d) > 32 unsupported", dk, dv);
  PG_REQUIRE(N <= 65535, PG_ESHAPE, "pg_attn_decode: N exceeds the grid limit");
  PG_REQUIRE(strict == 0 || strict == 1, PG_EINVAL, "pg_attn_decode: strict must be 0/1");
  const float scale2 = 1.44269504088896340736f / sqrtf((float)dk);
  dim3 grid((unsigned)heads, (unsigned)N);
  hipStream_t st = (hipStream_t)stream;
  if (dk <= 4 && dv <= 4)
    hipLaunchKernelGGL((attn_decode_kernel<4, 4>), grid, dim3(64), 0, st, qkv, k_cache, v_cache, o, N, heads, L, p,
                       strict, ld, dk, dv, scale2, pos_dev);
  else
    hipLaunchKernelGGL((attn_decode_kernel<32, 32>), grid, dim3(64), 0, st, qkv, k_cache, v_cache, o, N, heads, L,
                       p, strict, ld, dk, dv, scale2, pos_dev);
  PG_LAUNCH_CHECK("pg_attn_decode");
  return 0;
}
