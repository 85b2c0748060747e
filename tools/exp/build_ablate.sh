#!/bin/bash
# Timing ablations of attention_mfma.hip (results are WRONG by construction; timing only):
#   abl1: exp2 -> one multiply      abl2: 4x4x1 MFMA -> one FMA     abl3: both
#   abl4: 16x16x4 MFMA -> one FMA + splat
set -e
cd "$(dirname "$0")/../.."
SRC=pytorch-generative_amd/csrc/attention_mfma.hip
OBJ=pytorch-generative_amd/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Ipytorch-generative_amd/csrc"
others=$(ls $OBJ/*.o | grep -v attention_mfma.o)
for v in 1 2 3 4; do
  tmp=pytorch-generative_amd/csrc/_abl$v.hip
  cp $SRC $tmp
  if [ $((v & 1)) -ne 0 ] && [ $v -ne 4 ]; then
    sed -i 's|__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }|__device__ __forceinline__ float ex2(float x) { return x * 0.999f; }|' $tmp
  fi
  if [ $((v & 2)) -ne 0 ]; then
    sed -i 's|#define MFMA4(A, B, C) .*|__device__ __forceinline__ f32x4 abl4(float a, float b, f32x4 c) { c[0] = __builtin_fmaf(a, b, c[0]); return c; }\n#define MFMA4(A, B, C) abl4((A), (B), (C))|' $tmp
  fi
  if [ $v -eq 4 ]; then
    sed -i 's|#define MFMA16(A, B, C) .*|__device__ __forceinline__ f32x4 abl16(float a, float b, f32x4 c) { float t = a * b; return f32x4{c[0] + t, c[1] + t, c[2] - t, c[3] - t}; }\n#define MFMA16(A, B, C) abl16((A), (B), (C))|' $tmp
  fi
  /opt/rocm/bin/hipcc $FLAGS -c $tmp -o tools/exp/_abl$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others tools/exp/_abl$v.o -o tools/exp/libpg_abl$v.so
  rm -f $tmp tools/exp/_abl$v.o
done
ls -la tools/exp/libpg_abl*.so
