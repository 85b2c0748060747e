mkdir -p gpurun_out
{
for v in 1 0 1 0; do echo "DKV_BF16=$v"; PG_ATTN_DKV_BF16=$v timeout 300 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
