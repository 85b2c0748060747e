#!/bin/bash
# Phase ablation of the row-ring weight-gradient kernel (ablate build, WRONG results, timing only): PG_WB_DBG bits 1 no loads, 2 no commit, 4 no MFMA/reads
export PG_HIP_LIB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ablate.so
mkdir -p gpurun_out
for d in ${DBGS:-0 1 2 4 3 5 6 7}; do
  echo "PG_WB_DBG=$d"
  PG_WB_DBG=$d python tools/exp/wgrad_ab.py "snail 2x2 64->64 b1024" 2>/dev/null
done 2>&1 | tee gpurun_out/r06_ring_ablation.txt
