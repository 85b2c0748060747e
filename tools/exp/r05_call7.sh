#!/bin/bash
# Round 5: image-input convolutions (3 / 4 channels) on the fp32-MFMA kernel instead of the VALU tap kernel (PG_CONV_MFMA_MIN_CIN=3)
ulimit -c 0
OUT=gpurun_out/c7; mkdir -p $OUT
echo "== conv op tier + model tier with PG_CONV_MFMA_MIN_CIN=3"
PG_CONV_MFMA_MIN_CIN=3 timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_f4.py -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
echo "== throughput (images/s): default | PG_CONV_MFMA_MIN_CIN=3"
for m in beta_vae:1024 vd_vae:512 pixel_snail:1024 gated_pixel_cnn:512 pixel_cnn_pp:64; do
  M=${m%%:*}; B=${m##*:}
  a=$(timeout 150 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  b=$(PG_CONV_MFMA_MIN_CIN=3 timeout 150 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "$M  ${a#*: }  ${b#*: }"
done
