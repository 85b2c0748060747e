mkdir -p gpurun_out/stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/stats/*
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats -o s -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/stats/log.txt 2>&1
find $R/gpurun_out/stats -name "*kernel_trace.csv" -delete
find $R/gpurun_out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/stats/kernel_stats.csv
tail -1 $R/gpurun_out/stats/log.txt | cut -c1-200
