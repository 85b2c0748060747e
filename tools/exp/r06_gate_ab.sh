#!/bin/bash
# Fused GatedActivation epilogue (PG_FUSE_GATE) on PixelSNAIL, same box: parity tests first, then the bench both ways, twice.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -x -k "gated" -p no:cacheprovider > gpurun_out/r06_gate_tests.log 2>&1
tail -3 gpurun_out/r06_gate_tests.log
python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py -q -x -k "snail" -p no:cacheprovider >> gpurun_out/r06_gate_tests.log 2>&1
tail -3 gpurun_out/r06_gate_tests.log
for rep in 1 2; do
  for f in 1 0; do
    for b in 1024 64; do
      echo "PG_FUSE_GATE=$f batch=$b rep=$rep"
      PG_FUSE_GATE=$f python bench.py --model pixel_snail --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
    done
  done
done 2>&1 | tee gpurun_out/r06_gate_ab.txt
