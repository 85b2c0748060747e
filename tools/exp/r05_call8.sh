#!/bin/bash
# Round 5: one-channel input layers (PixelCNN 7x7 1->64, ImageGPT 3x3 1->16) on the fp32-MFMA kernel?
ulimit -c 0
OUT=gpurun_out/c8; mkdir -p $OUT
export PG_CONV_MFMA_MIN_CIN=1 PG_CONV_MFMA_MIN_COUT=16
echo "== op + model tier with PG_CONV_MFMA_MIN_CIN=1 PG_CONV_MFMA_MIN_COUT=16"
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
unset PG_CONV_MFMA_MIN_CIN PG_CONV_MFMA_MIN_COUT
echo "== throughput (images/s): default | MIN_CIN=1 MIN_COUT=32 | MIN_CIN=1 MIN_COUT=16"
for m in pixel_cnn:1024 image_gpt:1024 image_gpt:64; do
  M=${m%%:*}; B=${m##*:}
  a=$(timeout 150 python bench.py --model $M --batch $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  b=$(PG_CONV_MFMA_MIN_CIN=1 timeout 150 python bench.py --model $M --batch $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  c=$(PG_CONV_MFMA_MIN_CIN=1 PG_CONV_MFMA_MIN_COUT=16 timeout 150 python bench.py --model $M --batch $B --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "$M:$B  ${a#*: }  ${b#*: }  ${c#*: }"
done
