#!/bin/bash
# conv_b3r_kernel (row ring, register-resident weights) against conv_b3p_kernel on PixelSNAIL's 2x2 64->64: parity cases, launch times
# (forward under no_grad, forward + backward), then the model both ways (ab library, PG_CONV_B3R=0 = off)
mkdir -p gpurun_out
AB=$PWD/pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip_ab.so
python -m pytest tests/test_gpu_ops.py -q -k "test_conv" -p no:cacheprovider 2>&1 | tail -2
{
echo "== conv_b3r on (production)"; python tools/exp/q_ab.py "snail 2x2 64->64" 2>&1 | grep -v amdgpu.ids
echo "== off (ab library, PG_CONV_B3R=0)"; PG_HIP_LIB=$AB PG_CONV_B3R=0 python tools/exp/q_ab.py "snail 2x2 64->64" 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
  echo "on  pixel_snail: $(python bench.py --model pixel_snail --batch 1024 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  echo "off pixel_snail: $(PG_HIP_LIB=$AB PG_CONV_B3R=0 python bench.py --model pixel_snail --batch 1024 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
} | tee gpurun_out/r06_b3r_ab.txt
