mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
for cfg in 1 0 1 0; do echo "FUSE_BLOCK=$cfg"; PG_FUSE_BLOCK=$cfg timeout 300 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
