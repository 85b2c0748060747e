#!/bin/bash
# Round 5, GPU call 2: the big tiles of the bf16x3 weight gradient (conv_wgrad_b3_kernel<T, MR, 8, 4>): parity, launch A/B, end to end
ulimit -c 0
OUT=gpurun_out/c2; mkdir -p $OUT
echo "== conv op tier"
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" -p no:cacheprovider > $OUT/ops.log 2>&1
echo "rc=$? $(tail -1 $OUT/ops.log)"; grep -E "^(FAILED|ERROR)" $OUT/ops.log | head
echo "== launch A/B"
PG_WGRAD_B3_BIG=0 timeout 120 python tools/exp/wgrad_big_ab.py 2>/dev/null | tail -1
timeout 120 python tools/exp/wgrad_big_ab.py 2>/dev/null | tail -1
echo "== end to end (images/s): small tiles | big tiles"
for m in gated_pixel_cnn:512 pixel_snail:1024 pixel_cnn:1024; do
  M=${m%%:*}; B=${m##*:}
  a=$(PG_WGRAD_B3_BIG=0 timeout 120 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  b=$(timeout 120 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "$M  ${a#*: }  ${b#*: }"
done
