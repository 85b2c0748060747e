#!/bin/bash
# Row-ring weight gradient at the model level, same box: production library vs ab library with PG_WGRAD_B3_RING=0
mkdir -p gpurun_out
AB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
run() { python bench.py --model $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for m in "pixel_snail 1024" "gated_pixel_cnn 512" "pixel_cnn_pp 64"; do
    set -- $m
    echo "ring on  $1 b$2: $(run $1 $2)"
    echo "ring off $1 b$2: $(PG_HIP_LIB=$AB PG_WGRAD_B3_RING=0 run $1 $2)"
  done
done 2>&1 | tee gpurun_out/r06_ring_models.txt
