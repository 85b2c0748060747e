#!/bin/bash
# usage: bash tools/exp/prof_cmd.sh <tag> <cmd...>   (on the GPU box) -> gpurun_out/<tag>/kernel_stats.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- "$@" > $OUT/stats.log 2>&1
tail -20 $OUT/stats.log | cut -c1-300
f=$(find $OUT/stats -name "p_kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
find $OUT/stats -name "*.csv" -size +2M -delete
head -${PROF_LINES:-25} $OUT/kernel_stats.csv | cut -c1-220
