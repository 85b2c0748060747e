#!/bin/bash
# Round 5, GPU call 3: pipelined 4-tap kernel with several output chunks (PixelCNN++ widths) + where the wide kernel's time goes
ulimit -c 0
OUT=gpurun_out/c3; mkdir -p $OUT
echo "== conv op tier"
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv" -p no:cacheprovider > $OUT/ops.log 2>&1
echo "rc=$? $(tail -1 $OUT/ops.log)"; grep -E "^(FAILED|ERROR)" $OUT/ops.log | head
echo "== PixelCNN++ (images/s): one chunk only | several chunks on the pipelined kernel"
a=$(PG_CONV_B3P_MULTI=0 timeout 150 python bench.py --model pixel_cnn_pp --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
b=$(timeout 150 python bench.py --model pixel_cnn_pp --batch 64 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
echo "pixel_cnn_pp  ${a#*: }  ${b#*: }"
echo "== phase clocks of the convolution kernels on GatedPixelCNN's / PixelCNN++'s shapes"
timeout 200 python tools/exp/b3_phase_prof.py gated 2>&1 | tee $OUT/phase_clocks.txt | tail -90
