#!/bin/bash
# bench.py --dp-parity with two ranks on ONE GPU over gloo, several workloads: which of them reproduce the 1-rank run exactly?
export PG_FORCE_DEVICE=0 PG_DIST_BACKEND=gloo
mkdir -p gpurun_out/dp
for m in pixel_cnn:64 gated_pixel_cnn:16 pixel_snail:32 pixel_snail:32 image_gpt:64 vd_vae:8; do
  M=${m%%:*}; B=${m##*:}
  timeout 300 python bench.py --gpus 2 --dp-parity --model $M --steps 4 --warmup 1 --batch $B 2>gpurun_out/dp/$M.err | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['dp_parity']; print('$M', p['ok'], 'vs rank0', p['max_abs_diff_vs_rank0'], 'vs 1-rank', p['max_abs_diff_vs_one_rank_run'], 'pmax', round(p['param_abs_max'],3))"
done
