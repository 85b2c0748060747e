mkdir -p gpurun_out
{
for v in 0 1 2 3 4; do timeout 120 python tools/exp/run_ablate.py $v; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
