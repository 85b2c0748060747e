#!/bin/bash
export PG_FORCE_DEVICE=0 PG_DIST_BACKEND=gloo
mkdir -p gpurun_out/dp
AB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
for mask in 6 5 3 7; do
  PG_HIP_LIB=$AB PG_ATTN_MFMA_MASK=$mask timeout 300 python bench.py --gpus 2 --dp-parity --model pixel_snail --steps 4 --warmup 1 --batch 32 2>gpurun_out/dp/x.err | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['dp_parity']; print('mask $mask', p['ok'], 'vs 1-rank', p['max_abs_diff_vs_one_rank_run'])" || { echo "mask $mask FAILED"; tail -2 gpurun_out/dp/x.err; }
done
