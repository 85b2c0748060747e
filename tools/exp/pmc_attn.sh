#!/bin/bash
# PMC passes over tools/attn_kernels.py (runs ON THE GPU BOX). usage: bash tools/exp/pmc_attn.sh [tag]
TAG=${1:-m44}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/attn_kernels.py 1024 5"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o x -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_ANY --output-format csv -d $OUT/p2 -o x -- $CMD > $OUT/p2.log 2>&1
python - <<PY
import csv, collections, json, os, glob, re
out = "$OUT"
res = collections.defaultdict(dict)
for tag in ("p1", "p2"):
    for path in glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(path)):
            a = agg[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, d in agg.items():
            if "attn" in k:
                for c, v in d.items():
                    res[re.search(r"attn_\w+", k).group(0)][c] = v[0] / max(v[1], 1)
        os.remove(path)
    for p in glob.glob(os.path.join(out, tag, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(p)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
tail -3 $OUT/p1.log
