#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the
# HBM counters (never combined with other trace domains), outputs under gpurun_out/prof_r01/.
# usage: bash tools/collect_profiles.sh [batch]
B=${1:-1024}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r01 -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o r01 -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o r01 -- $CMD > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/sq -o r01 -- $CMD > $OUT/sq.log 2>&1
# keep only small summaries (counter CSVs are aggregated here to stay under the merge limit)
python - <<PY
import csv, collections, json, os
out = "$OUT"
res = {}
for tag in ("fetch", "write", "sq"):
    path = os.path.join(out, tag, "r01_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        os.remove(path)
    res[tag] = {k: {c: {"sum": v[0], "dispatches": v[1]} for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for tag in ("fetch", "write", "sq", "stats"):
    p = os.path.join(out, tag, "r01_kernel_trace.csv")
    if os.path.exists(p): os.remove(p)
PY
tail -1 $OUT/stats.log | cut -c1-200
ls -R $OUT | head -30
