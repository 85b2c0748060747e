mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv" 2>&1 | tail -2
for m in pixel_snail gated_pixel_cnn; do echo "model $m"; timeout 400 python bench.py --model $m --batch 128 --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
