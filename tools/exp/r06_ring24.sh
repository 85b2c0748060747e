#!/bin/bash
# one-tap weight gradients: the 128 x 128 ring tile (PG_WGRAD_B3_RING_CFG=24, ab library) against the big-tile kernel (default routing)
mkdir -p gpurun_out
export PG_HIP_LIB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
{
for cfg in "" 24; do
  echo "== PG_WGRAD_B3_RING_CFG=$cfg"
  PG_WGRAD_B3_RING_CFG=$cfg python tools/exp/wgrad_ab.py "1x1" 2>&1 | grep -v amdgpu.ids
done
} | tee gpurun_out/r06_ring24.txt
