#!/bin/bash
# Gate epilogue for 2C = 256 channels with the convolution's own residual (GatedPixelCNN): parity tests, then PG_FUSE_GATE=1 / 0, twice
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "gate" -p no:cacheprovider 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py tests/test_gpu_shared_device.py -q -x -k "gated" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
  for f in 1 0; do
    echo "PG_FUSE_GATE=$f rep=$rep: $(PG_FUSE_GATE=$f python bench.py --model gated_pixel_cnn --batch 512 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  done
done 2>&1 | tee gpurun_out/r06_gate256_ab.txt
