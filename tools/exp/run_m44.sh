mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
bash tools/collect_profiles.sh 1024 > gpurun_out/collect.log 2>&1
cd $R
{
python bench.py > gpurun_out/bench_default.json
tail -1 gpurun_out/bench_default.json | cut -c1-300
python __graft_entry__.py --smoke 2>&1 | tail -1
} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/run.log
