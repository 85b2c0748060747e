#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round-4 measurement record.
#   1. rocprofv3 --kernel-trace --stats of bench.py for each of the six workloads -> gpurun_out/prof_r04/<model>_kernel_stats.csv
#   2. SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters; never combined with other trace domains) over a few
#      launches of PixelSNAIL's dominant convolution at the bench's batch (tools/exp/conv_launch.py), with ops.add on the
#      same tensor size as the calibration kernel
# usage: bash tools/collect_profiles_r04.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for spec in image_gpt:1024 pixel_snail:1024 pixel_cnn:1024 gated_pixel_cnn:512 beta_vae:1024 vd_vae:512; do
  M=${spec%%:*}; B=${spec##*:}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${M}_stats -o p -- \
    python $R/bench.py --model $M --steps 10 --warmup 3 --batch $B --no-cpu-baseline > $OUT/${M}_stats.log 2>&1 || echo "[$M] rc=$?"
  f=$(find $OUT/${M}_stats -name "p_kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${M}_kernel_stats.csv
  rm -rf $OUT/${M}_stats
  grep -o '"value": [0-9.]*' $OUT/${M}_stats.log | head -1
done
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for pass in FETCH_SIZE WRITE_SIZE "$SQ"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 240 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/conv_$tag -o p -- \
    python $R/tools/exp/conv_launch.py 1024 5 > $OUT/conv_$tag.log 2>&1 || echo "[conv $tag] rc=$?"
done
python - <<PY
import csv, collections, glob, json, os
out = "$OUT"
res = {}
for d in glob.glob(os.path.join(out, "conv_*")):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            a = agg[r["Kernel_Name"][:100]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, dd in agg.items():
        res.setdefault(k, {}).update({c: {"per_dispatch": v[0] / max(v[1], 1), "dispatches": v[1]} for c, v in dd.items()})
json.dump(res, open(os.path.join(out, "conv_pmc.json"), "w"), indent=1)
for k, v in res.items():
    if "conv_b3" in k or "add" in k:
        print(k[:70], {c: round(x["per_dispatch"]) for c, x in v.items()})
PY
rm -rf $OUT/conv_FETCH_SIZE $OUT/conv_WRITE_SIZE $OUT/conv_SQ_WAVE_CYCLES
ls $OUT
