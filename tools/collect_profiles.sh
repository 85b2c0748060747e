#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + SEPARATE PMC passes for the HBM and
# SQ counters (never combined with other trace domains) for both halves of the bench line, outputs
# under gpurun_out/prof_r02/.   usage: bash tools/collect_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
for W in "igpt --model image_gpt --batch 1024" "snail --model pixel_snail --batch 512"; do
  set -- $W; TAG=$1; shift
  CMD="python $R/bench.py $* --steps 10 --warmup 3 --no-cpu-baseline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -o p -- $CMD > $OUT/${TAG}_stats.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o p -- $CMD > $OUT/${TAG}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o p -- $CMD > $OUT/${TAG}_write.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/${TAG}_sq -o p -- $CMD > $OUT/${TAG}_sq.log 2>&1
  tail -1 $OUT/${TAG}_stats.log | cut -c1-160
done
# keep only small summaries (counter CSVs are aggregated here to stay under the merge limit)
python - <<PY
import csv, collections, glob, json, os
out = "$OUT"
res = {}
for tag in ("igpt", "snail"):
    for kind in ("fetch", "write", "sq"):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for path in glob.glob(os.path.join(out, f"{tag}_{kind}", "**", "p_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                a = agg[r["Kernel_Name"]][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
            os.remove(path)
        res[f"{tag}_{kind}"] = {k: {c: {"sum": v[0], "dispatches": v[1]} for c, v in d.items()} for k, d in agg.items()}
    for f in glob.glob(os.path.join(out, f"{tag}_stats", "**", "p_kernel_stats.csv"), recursive=True):
        os.replace(f, os.path.join(out, f"{tag}_kernel_stats.csv"))
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(out, "**", "*.csv"), recursive=True):
    if os.path.getsize(f) > (1 << 20):
        os.remove(f)
PY
ls $OUT | head -30
