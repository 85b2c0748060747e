#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of one bench.py workload.
# usage: bash tools/prof_model.sh <model> <batch> <tag>
M=${1:-pixel_snail}; B=${2:-128}; TAG=${3:-r02_$M}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --model $M --steps 10 --warmup 3 --batch $B --no-cpu-baseline ${PROF_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- $CMD > $OUT/stats.log 2>&1
tail -2 $OUT/stats.log | cut -c1-600
f=$(find $OUT/stats -name "p_kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
find $OUT/stats -name "*.csv" -size +2M -delete
head -40 $OUT/kernel_stats.csv | cut -c1-200
