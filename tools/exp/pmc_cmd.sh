#!/bin/bash
# usage: bash tools/exp/pmc_cmd.sh <tag> "<counters>" <cmd...>  -> gpurun_out/<tag>/pmc.json (per-kernel sums)
TAG=$1; CNT=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT/pmc -o p -- "$@" > $OUT/pmc.log 2>&1
python - <<PY
import csv, collections, json, glob, os
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob("$OUT/pmc/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        a = agg[r["Kernel_Name"][:90]][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    os.remove(path)
res = {k: {c: {"per_dispatch": v[0] / max(v[1], 1), "dispatches": v[1]} for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open("$OUT/pmc.json", "w"), indent=1)
for k, d in res.items():
    if "${PMC_FILTER:-}" in k:
        print(k[:80]); [print("   ", c, round(v["per_dispatch"]), v["dispatches"]) for c, v in d.items()]
PY
find $OUT/pmc -name "*.csv" -size +1M -delete
