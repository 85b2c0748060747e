#!/bin/bash
# First GPU call of the next round (run through gpurun from the repo root, ~2 GPU-minutes):
# validates the two pieces that were written after round 2's GPU budget was spent.
#   gpurun --timeout 600 -- 'bash tools/exp/next_round.sh'
ulimit -c 0
echo "== 1. wide conv_b3 workgroups (PG_CONV_B3_WIDE=1): values against the VALU kernels + time"
echo "-- default"
timeout 120 python tools/exp/conv_ab.py gated "snail 2x2 64->128" 2>&1 | tail -4
echo "-- wide"
PG_CONV_B3_WIDE=1 timeout 120 python tools/exp/conv_ab.py gated "snail 2x2 64->128" 2>&1 | tail -4
echo "-- vector epilogue"
PG_B3_VEC_EP=1 timeout 120 python tools/exp/conv_ab.py gated snail 2>&1 | tail -5
echo "-- wide + vector epilogue"
PG_CONV_B3_WIDE=1 PG_B3_VEC_EP=1 timeout 120 python tools/exp/conv_ab.py gated "snail 2x2 64->128" 2>&1 | tail -4
echo "== 2. f4 (VectorQuantizer / VQ-VAE / VQ-VAE-2) against the reference goldens"
PG_TEST_F4=1 timeout 200 python -m pytest tests/test_gpu_f4.py -m gpu -q 2>&1 | tail -15
echo "== 3. if 1. is correct and faster: whole models with the wide kernels"
for m in gated_pixel_cnn pixel_snail; do
  for w in 0 1; do
    PG_CONV_B3_WIDE=$w timeout 150 python bench.py --model $m --batch 512 --steps 10 --warmup 3 --no-extras \
      --no-cpu-baseline 2>&1 | tail -1 | cut -c1-160
  done
done
