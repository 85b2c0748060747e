#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round-6 measurement record.
#   bash tools/collect_profiles_r06.sh pmc     -> counter passes (separate --pmc runs, never combined with other trace domains)
#                                                 over tools/exp/pmc_launch.py: FETCH_SIZE | WRITE_SIZE | SQ counters
#   bash tools/collect_profiles_r06.sh stats   -> rocprofv3 --kernel-trace --stats of bench.py for every workload
# Output under gpurun_out/prof_r06/; tools/summarize_profiles_r06.py turns it into the tracked files of profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=${1:-pmc}
if [ "$MODE" = "stats" ]; then
  for spec in ${SPECS:-image_gpt:1024 image_gpt:64 pixel_snail:1024 pixel_cnn:1024 gated_pixel_cnn:512 pixel_cnn_pp:64 beta_vae:1024 vd_vae:512}; do
    M=${spec%%:*}; B=${spec##*:}; TAG=$M; [ "$spec" = "image_gpt:64" ] && TAG=image_gpt_b64
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o p -- \
      python $R/bench.py --model $M --steps 10 --warmup 3 --batch $B --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1 || echo "[$TAG] rc=$?"
    f=$(find $OUT/${TAG}_stats -name "p_kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv
    rm -rf $OUT/${TAG}_stats
    echo "$TAG $(grep -o '"value": [0-9.]*' $OUT/${TAG}_stats.log | head -1)"
  done
  exit 0
fi
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for pass in FETCH_SIZE WRITE_SIZE "$SQ"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$tag -o p -- \
    python $R/tools/exp/pmc_launch.py 3 --wide --q > $OUT/pmc_$tag.log 2>&1 || echo "[pmc $tag] rc=$?"
  tail -1 $OUT/pmc_$tag.log
done
python - <<PY
import csv, collections, glob, json, os
out = "$OUT"
res = {}
for d in glob.glob(os.path.join(out, "pmc_*")):
    if not os.path.isdir(d):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, "**", "p_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            a = agg[r["Kernel_Name"][:140]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, dd in agg.items():
        res.setdefault(k, {}).update({c: {"per_dispatch": v[0] / max(v[1], 1), "dispatches": v[1]} for c, v in dd.items()})
json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1)
for k, v in res.items():
    if any(t in k for t in ("attn_", "conv_b3", "conv_wgrad", "add_kernel", "wgrad_reduce")):
        print(k[:90], {c: round(x["per_dispatch"]) for c, x in v.items()})
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_WAVE_CYCLES
ls $OUT
