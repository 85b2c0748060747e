mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
bash tools/collect_profiles.sh 1024 > gpurun_out/collect.log 2>&1
cd $R
{
python bench.py > gpurun_out/bench_default.json
tail -1 gpurun_out/bench_default.json | cut -c1-400
for b in 64 512 2048; do echo "batch $b"; timeout 300 python bench.py --batch $b --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
for m in pixel_cnn gated_pixel_cnn pixel_snail beta_vae vd_vae; do echo "model $m"; timeout 400 python bench.py --model $m --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config'].get('per_gpu_batch'))"; done
python __graft_entry__.py --smoke 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
