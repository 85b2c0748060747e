#!/bin/bash
# Dual data gradient of PixelSNAIL's block tail (PG_FUSE_DUAL): parity tests, then the bench both ways, twice.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -k "dual or gated or protocol" -p no:cacheprovider > gpurun_out/r06_dual_tests.log 2>&1
tail -3 gpurun_out/r06_dual_tests.log
python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py tests/test_gpu_shared_device.py -q -x -k "snail" -p no:cacheprovider >> gpurun_out/r06_dual_tests.log 2>&1
tail -3 gpurun_out/r06_dual_tests.log
for rep in 1 2; do
  for f in 1 0; do
    for b in 1024 128; do
      echo "PG_FUSE_DUAL=$f batch=$b rep=$rep: $(PG_FUSE_DUAL=$f python bench.py --model pixel_snail --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
    done
  done
done 2>&1 | tee gpurun_out/r06_dual_ab.txt
