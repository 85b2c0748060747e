mkdir -p gpurun_out
{
for v in cur sonly cur sonly; do
  lib=pytorch-generative_amd/pytorch_generative_amd/lib/libpg_hip.so; [ $v = sonly ] && lib=tools/exp/libpg_sonly.so
  echo "lib $v"; timeout 300 python tools/exp/bench_with_lib.py $lib --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/run.log
