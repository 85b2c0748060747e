mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
{
for cfg in "2048,512,2048,512" "1280,768,1024,512" "1024,1024,2048,512" "2560,768,3072,512"; do
  rm -rf /tmp/st; PG_BLOCK_GRID=$cfg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -o s -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/st.log 2>&1
  echo "cfg $cfg"; python $R/tools/exp/parse_stats.py /tmp/st
done
} 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/run.log
