#!/bin/bash
# the two-tap 256 -> 256: 256 x 64 tiles (one workgroup per CU, default) against 128 x 64 tiles at two workgroups per CU (PG_WGRAD_B3_RING_CFG=22)
mkdir -p gpurun_out
export PG_HIP_LIB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
{
for cfg in "" 22; do
  echo "== PG_WGRAD_B3_RING_CFG=$cfg"
  PG_WGRAD_B3_RING_CFG=$cfg python tools/exp/wgrad_ab.py "gated 2x1" "gated 1x2" "snail 2x2 64->128" 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_ring22.txt
