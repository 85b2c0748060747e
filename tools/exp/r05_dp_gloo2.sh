#!/bin/bash
# Which kernel makes PixelSNAIL's two-process run differ from its one-process run? (bench.py --dp-parity, two ranks on one GPU over gloo)
export PG_FORCE_DEVICE=0 PG_DIST_BACKEND=gloo
mkdir -p gpurun_out/dp
one() {  # label, model, batch, env...
  local label=$1 M=$2 B=$3; shift 3
  env "$@" timeout 300 python bench.py --gpus 2 --dp-parity --model $M --steps 4 --warmup 1 --batch $B 2>gpurun_out/dp/x.err | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['dp_parity']; print('$label', p['ok'], 'vs rank0', p['max_abs_diff_vs_rank0'], 'vs 1-rank', p['max_abs_diff_vs_one_rank_run'])" || { echo "$label FAILED"; tail -2 gpurun_out/dp/x.err; }
}
one beta_vae beta_vae 16 A=1
one pixel_cnn_pp pixel_cnn_pp 4 A=1
AB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
one "snail ab-default" pixel_snail 32 PG_HIP_LIB=$AB
one "snail PG_CONV_B3P=0" pixel_snail 32 PG_HIP_LIB=$AB PG_CONV_B3P=0
one "snail PG_CONV_B3=0" pixel_snail 32 PG_HIP_LIB=$AB PG_CONV_B3=0
one "snail PG_ATTN_MFMA=0" pixel_snail 32 PG_HIP_LIB=$AB PG_ATTN_MFMA=0
one "snail PG_FUSE_QKV_EXTRA=0" pixel_snail 32 PG_FUSE_QKV_EXTRA=0
one "snail PG_FUSE_SKIP=0" pixel_snail 32 PG_FUSE_SKIP=0
