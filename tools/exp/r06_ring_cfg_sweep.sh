#!/bin/bash
# Wave-tile sweep of the row-ring weight-gradient kernel (ab library): default routing, forced tiles, ring off.
mkdir -p gpurun_out
export PG_HIP_LIB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
{
for cfg in "" 12 22 44 42; do
  echo "== PG_WGRAD_B3_RING_CFG=$cfg"
  PG_WGRAD_B3_RING_CFG=$cfg python tools/exp/wgrad_ab.py "$@" 2>&1 | grep -v amdgpu.ids
done
echo "== ring off"
PG_WGRAD_B3_RING=0 python tools/exp/wgrad_ab.py "$@" 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r06_ring_cfg_sweep.txt
