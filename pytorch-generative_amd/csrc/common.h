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

__device__ __forceinline__ float pg_apply_act(float x, int act) {
  switch (act) {
    case PG_ACT_RELU:
      return x > 0.f ? x : 0.f;
    case PG_ACT_ELU:
      return x > 0.f ? x : expm1f(x);
    case PG_ACT_GELU:
      return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
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
      return x > 0.f ? 1.f : expf(x);
    case PG_ACT_GELU: {
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
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
