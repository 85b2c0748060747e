#!/bin/bash
export PG_FORCE_DEVICE=0 PG_DIST_BACKEND=gloo
mkdir -p gpurun_out/dp
for extra in "" "--no-graph"; do
  timeout 300 python bench.py --gpus 2 --dp-parity --model pixel_snail --steps 4 --warmup 1 --batch 32 $extra 2>gpurun_out/dp/x.err | grep "^{" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['dp_parity']; print('[$extra]', d['config']['launch'], p['ok'], 'vs 1-rank', p['max_abs_diff_vs_one_rank_run'])" || { echo "[$extra] FAILED"; tail -2 gpurun_out/dp/x.err; }
done
