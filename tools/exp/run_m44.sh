mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention or attn or block" 2>&1 | tail -3
for v in 1 2 3; do timeout 120 python tools/attn_kernels.py 1024; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
