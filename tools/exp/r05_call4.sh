#!/bin/bash
# Round 5, GPU call 4: weight slabs of conv_b3_kernel by LDS-DMA (global_load_lds_dwordx4), variant library against production
ulimit -c 0
L=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib
OUT=gpurun_out/c4; mkdir -p $OUT
V=${1:-wglds}
echo "== op + model tier with the $V library"
PG_HIP_LIB=$L/libpg_hip_$V.so timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_$V.log 2>&1
echo "rc=$? $(tail -1 $OUT/tests_$V.log)"; grep -E "^(FAILED|ERROR)" $OUT/tests_$V.log | head
echo "== throughput, same box (images/s): prod | $V"
for m in gated_pixel_cnn:512 pixel_snail:1024 pixel_cnn:1024 vd_vae:512 beta_vae:1024 pixel_cnn_pp:64; do
  M=${m%%:*}; B=${m##*:}
  line="$M"
  for v in prod $V; do
    so=$L/libpg_hip_$v.so; [ $v = prod ] && so=$L/libpg_hip.so
    r=$(timeout 150 python tools/exp/bench_with_lib.py $so --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
    line="$line  $v ${r#*: }"
  done
  echo "$line"
done
