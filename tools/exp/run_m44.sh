mkdir -p gpurun_out
{
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
python __graft_entry__.py --smoke 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
