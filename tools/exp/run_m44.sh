mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
