// common.h — shared helpers for the gfx950 kernels of libpg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pg_hip.h"

#define PG_EXPORT extern "C" __attribute__((visibility("default")))

void pg_set_error(const char* fmt, ...);

#define PG_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) {                  \
      pg_set_error(__VA_ARGS__);    \
      return (code);                \
    }                               \
  } while (0)

// Check the launch that was just enqueued (no sync: capturable).
#define PG_LAUNCH_CHECK(name)                                           \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) {                                            \
      pg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                  \
    }                                                                   \
  } while (0)

static inline int pg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The two pieces of the exact (erf) GELU, branch-free: cdf = Phi(x) = (1 + erf(x / sqrt 2)) / 2 and
// e = exp(-x^2 / 2), the exponential the density shares with erf's tail. erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 absolute: one fp32 rounding of 1 + erf); for x < 0 the tail 0.5 (1 - erf) is used as it is,
// without forming 1 + erf(negative). Against float64 the fp32 results are as close as ATen's own fp32 GELU
// (max abs error 4.7e-7 vs 1.2e-6 on [-8, 8], derivative 3.2e-7 vs 2.9e-7; tests/test_cpu_gelu.py restates this
// arithmetic). libm's erff is ~40 instructions with a divergent branch per element — it split the staging loops of
// the convolutions and the MLP chain of the ImageGPT block kernels into one basic block per element.
__device__ __forceinline__ void pg_gelu_parts(float x, float& cdf, float& e) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  e = __expf(-z * z);
  const float q = 0.5f * p * e;  // Phi(-|x|)
  cdf = x >= 0.f ? 1.f - q : q;
}
__device__ __forceinline__ float pg_gelu(float x) {
  float cdf, e;
  pg_gelu_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float pg_gelu_grad(float x) {
  float cdf, e;
  pg_gelu_parts(x, cdf, e);
  return fmaf(x * 0.39894228040143267794f, e, cdf);
}

__device__ __forceinline__ float pg_apply_act(float x, int act) {
  switch (act) {
    case PG_ACT_RELU:
      return x > 0.f ? x : 0.f;
    case PG_ACT_ELU:
      // exp(x) - 1 on the hardware exp2 path (absolute error <= 1.2e-7 next to activations of order 1;
      // libm's expm1f is ~25 instructions per element in the staging loops that apply this)
      return x > 0.f ? x : __expf(x) - 1.0f;
    case PG_ACT_GELU:
      return pg_gelu(x);
    default:
      return x;
  }
}

// d act(x) / dx
__device__ __forceinline__ float pg_act_grad(float x, int act) {
  switch (act) {
    case PG_ACT_RELU:
      return x > 0.f ? 1.f : 0.f;
    case PG_ACT_ELU:
      return x > 0.f ? 1.f : __expf(x);  // hardware exp2 path, as the forward (x <= 0: relative error ~1e-7)
    case PG_ACT_GELU:
      return pg_gelu_grad(x);
    case PG_ACT_ELU_OUT:  // x is ELU's OUTPUT: elu'(pre) = y > 0 ? 1 : y + 1
      return x > 0.f ? 1.f : x + 1.f;
    default:
      return 1.f;
  }
}

// d act / d pre-activation expressed through the activation's output y (ReLU, ELU only)
__device__ __forceinline__ float pg_act_grad_out(float y, int act) {
  switch (act) {
    case PG_ACT_RELU:
      return y > 0.f ? 1.f : 0.f;
    case PG_ACT_ELU:
      return y > 0.f ? 1.f : y + 1.f;
    default:
      return 1.f;
  }
}

// 64-lane wavefront sum (all lanes get the result).
__device__ __forceinline__ float pg_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Stage `rows` = nch * tile_h rows of `tile_w` columns from channel planes in global memory into an
// LDS tile laid out [ch][tile_h][tile_w] (channel stride ch_stride), zero filling everything outside
// the plane and applying the (compile-time) activation once. Row r of channel c maps to input row
// in_row0 + r and tile column t to input column t + min_dc.
// The loop is pure latency, so it is written with NO branches at all — clamped unconditional loads,
// a select, and an unconditional store whose address is redirected to a `dump` word for lanes that
// have nothing to write (hipcc otherwise sinks each load into its predicated store: load, wait,
// store, one at a time) — and unrolled 8x: 8 independent loads per lane are in flight before the
// first wait. `rpi` rows are packed per wave iteration (lanes = 1 << swp_shift columns).
template <int ACT>
__device__ __forceinline__ void pg_stage_rows(float* __restrict__ lds, int dump, int ch_stride,
                                              int tile_h, int tile_w, float inv_tile_h,
                                              const float* __restrict__ src, size_t plane_stride,
                                              int IH, int IW, int nch, int in_row0, int min_dc,
                                              int swp_shift, int rpi, int wave, int nwaves, int lane) {
  const int sub = lane >> swp_shift, col = lane & ((1 << swp_shift) - 1);
  const int rows = nch * tile_h;
  const int step = nwaves * rpi;
  const int iters = (rows + step - 1) / step;
  const int wcols = 1 << swp_shift;
  for (int tc0 = 0; tc0 < tile_w; tc0 += wcols) {  // one pass unless the tile is wider than 64
    const int tc = tc0 + col;
    const int ic = tc + min_dc;
    const int icc = ic < 0 ? 0 : (ic >= IW ? IW - 1 : ic);
    const bool cok = ic >= 0 && ic < IW;
    const bool cwr = tc < tile_w;
    for (int it0 = 0; it0 < iters; it0 += 8) {
      float v[8];
      int didx[8];
      bool ok[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {  // phase 1: eight independent loads
        const int rr = (it0 + k) * step + wave * rpi + sub;
        const bool rvalid = rr < rows;
        const int rrc = rvalid ? rr : rows - 1;
        const int ch = (int)(((float)rrc + 0.5f) * inv_tile_h);
        const int tr = rrc - ch * tile_h;
        const int ir = in_row0 + tr;
        const int irc = ir < 0 ? 0 : (ir >= IH ? IH - 1 : ir);
        v[k] = src[(size_t)ch * plane_stride + (size_t)irc * IW + icc];
        ok[k] = cok && ir >= 0 && ir < IH;
        didx[k] = (rvalid && cwr) ? ch * ch_stride + tr * tile_w + tc : dump;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)  // phase 2: eight LDS stores
        lds[didx[k]] = ok[k] ? pg_apply_act(v[k], ACT) : 0.f;
    }
  }
}

// float4 variant of pg_stage_rows for planes whose rows are 16-byte aligned (IW % 4 == 0, aligned
// base): a lane loads 4 consecutive input columns with one global_load_dwordx4. Only in-range
// elements are written, so the caller zero-fills the tile ONCE (halo columns / out-of-range rows
// are never written afterwards and stay zero). `qshift`: lanes per row = 1 << qshift >= IW/4.
template <int ACT>
__device__ __forceinline__ void pg_stage_rows_vec4(float* __restrict__ lds, int dump, int ch_stride,
                                                   int tile_h, int tile_w, float inv_tile_h,
                                                   const float* __restrict__ src,
                                                   size_t plane_stride, int IH, int IW, int nch,
                                                   int in_row0, int min_dc, int qshift, int wave,
                                                   int nwaves, int lane) {
  const int rpi = 64 >> qshift;
  const int sub = lane >> qshift, q = lane & ((1 << qshift) - 1);
  const int rows = nch * tile_h;
  const int step = nwaves * rpi;
  const int iters = (rows + step - 1) / step;
  const bool qok = 4 * q < IW;
  const int qc = qok ? q : 0;
  const int tcol = 4 * q - min_dc;  // tile column of the quad's first element
  for (int it0 = 0; it0 < iters; it0 += 8) {
    float4 v[8];
    int didx[8];
    bool vok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int rr = (it0 + k) * step + wave * rpi + sub;
      const bool rvalid = rr < rows;
      const int rrc = rvalid ? rr : rows - 1;
      const int ch = (int)(((float)rrc + 0.5f) * inv_tile_h);
      const int tr = rrc - ch * tile_h;
      const int ir = in_row0 + tr;
      vok[k] = rvalid && qok && ir >= 0 && ir < IH;
      const int irc = ir < 0 ? 0 : (ir >= IH ? IH - 1 : ir);
      v[k] = *reinterpret_cast<const float4*>(src + (size_t)ch * plane_stride + (size_t)irc * IW + 4 * qc);
      didx[k] = ch * ch_stride + tr * tile_w + tcol;  // may be -1..-3 for the first quad: per-element check below
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = vok[k];
      const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // a quad may straddle the tile's column range when no tap reaches its outer columns
        const bool oki = ok && (tcol + i) >= 0 && (tcol + i) < tile_w;
        lds[oki ? didx[k] + i : dump] = pg_apply_act(e[i], ACT);
      }
    }
  }
}

// Phase-ablation switches of the bf16x3 kernels (PG_B3_DBG / PG_WB_DBG: skip loads / commit / MFMA / epilogue —
// WRONG results, timing only) exist only in builds made with -DPG_ABLATE (`PG_ABLATE=1 python build.py`, which
// writes lib/libpg_hip_ablate.so for tools/exp); in the production library the predicate is a compile-time false
// and the environment variables are not read.
#ifdef PG_ABLATE
#define PG_DBG_BIT(flags, bit) (((flags) & (bit)) != 0)
#else
#define PG_DBG_BIT(flags, bit) false
#endif

// A/B switches of the kernels' host code ("A/B switches"): measurement only, every default is the fast
// path. The PRODUCTION library does not read them — PG_AB_ENV is a compile-time null there, so no environment variable can
// change which kernel a launch takes; `PG_VARIANT=ab python build.py` (-DPG_AB) builds lib/libpg_hip_ab.so with the
// switches live, for tools/exp (PG_HIP_LIB=<that file>). PG_CONV_LOG (a diagnostic: one stderr line per launch) stays live.
#ifdef PG_AB
#define PG_AB_ENV(name) getenv(name)
#else
#define PG_AB_ENV(name) ((const char*)nullptr)
#endif
