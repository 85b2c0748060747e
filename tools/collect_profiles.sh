#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + SEPARATE PMC passes for the HBM and
# SQ counters (never combined with other trace domains), outputs under gpurun_out/prof_r03/.
#   ImageGPT headline (batch 1024, hipGraph replay): stats, FETCH_SIZE, WRITE_SIZE, SQ counters.
#   PixelSNAIL (batch 512): stats on the graphed run; the counter passes on an UNGRAPHED 2-step run
#   restricted to the matrix-core kernels (counter collection over the ~11 k-launch graph stalled in
#   round 2: 25 GPU-minutes lost).
# usage: bash tools/collect_profiles.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"
run() {  # tag, timeout, rocprof args..., -- cmd
  local tag=$1 to=$2; shift 2
  timeout $to rocprofv3 "$@" > $OUT/$tag.log 2>&1 || echo "[$tag] rc=$?"
}
IG="python $R/bench.py --model image_gpt --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline"
run igpt_stats 300 --kernel-trace --stats --output-format csv -d $OUT/igpt_stats -o p -- $IG
run igpt_fetch 300 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/igpt_fetch -o p -- $IG
run igpt_write 300 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/igpt_write -o p -- $IG
run igpt_sq 300 --kernel-trace --pmc $SQ --output-format csv -d $OUT/igpt_sq -o p -- $IG
tail -1 $OUT/igpt_stats.log | cut -c1-160
SN="python $R/bench.py --model pixel_snail --batch 512 --steps 10 --warmup 3 --no-cpu-baseline"
run snail_stats 300 --kernel-trace --stats --output-format csv -d $OUT/snail_stats -o p -- $SN
SNE="python $R/bench.py --model pixel_snail --batch 512 --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
RX='conv_b3|conv_wgrad_b3|attn_.*_k4|gated_fwd4'
run snail_fetch 240 --kernel-trace --kernel-include-regex "$RX" --pmc FETCH_SIZE --output-format csv -d $OUT/snail_fetch -o p -- $SNE
run snail_write 240 --kernel-trace --kernel-include-regex "$RX" --pmc WRITE_SIZE --output-format csv -d $OUT/snail_write -o p -- $SNE
run snail_sq 240 --kernel-trace --kernel-include-regex "$RX" --pmc $SQ --output-format csv -d $OUT/snail_sq -o p -- $SNE
tail -1 $OUT/snail_stats.log | cut -c1-160
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
