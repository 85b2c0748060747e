#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel tables taken AFTER the last kernel changes of round 4 (the fused attention backward without its
# memset node, the bigger staged tiles) for the workloads they touch, plus PixelCNN++ (first profile). Same command form as
# tools/collect_profiles_r04.sh.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in ${PG_PROF_SPECS:-pixel_snail:1024 pixel_cnn_pp:64 vd_vae:512}; do
  M=${spec%%:*}; B=${spec##*:}
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${M}_stats -o p -- \
    python $R/bench.py --model $M --steps 10 --warmup 3 --batch $B --no-cpu-baseline > $OUT/${M}_stats.log 2>&1 || echo "[$M] rc=$?"
  f=$(find $OUT/${M}_stats -name "p_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${M}_kernel_stats.csv
  rm -rf $OUT/${M}_stats
  grep -o '"value": [0-9.]*' $OUT/${M}_stats.log | head -1
done
ls $OUT
