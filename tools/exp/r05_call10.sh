#!/bin/bash
# Round 5: bf16x3 convolution kernels for images of 64-255 pixels (PG_CONV_B3_MIN_PX=64, ab library) instead of the fp32-MFMA kernel?
ulimit -c 0
L=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib
OUT=gpurun_out/c10; mkdir -p $OUT
export PG_HIP_LIB=$L/libpg_hip_ab.so
echo "== op + model + f4 tier, ab library, PG_CONV_B3_MIN_PX=64"
PG_CONV_B3_MIN_PX=64 timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_f4.py -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests.log)"; grep -E "^(FAILED|ERROR)" $OUT/tests.log | head
echo "== throughput (images/s), ab library: 256 | 64 | 16"
for m in pixel_cnn_pp:64 vd_vae:512 beta_vae:1024; do
  M=${m%%:*}; B=${m##*:}
  line="$M"
  for px in 256 64 16; do
    r=$(PG_CONV_B3_MIN_PX=$px timeout 150 python bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    line="$line  ${r#*: }"
  done
  echo "$line"
done
