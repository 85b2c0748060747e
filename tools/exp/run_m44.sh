mkdir -p gpurun_out
{
PG_ATTN_DKV_BF16=1 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention or attn or block" 2>&1 | tail -3
for v in 0 1 0 1; do echo "DKV_BF16=$v"; PG_ATTN_DKV_BF16=$v timeout 120 python tools/attn_kernels.py 1024; done
echo "bf16 W=4"; PG_ATTN_DKV_BF16=1 PG_ATTN_WAVES=4,8,4 timeout 120 python tools/attn_kernels.py 1024
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
