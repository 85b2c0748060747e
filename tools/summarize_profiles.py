"""Turns gpurun_out/prof_r01/ (made by tools/collect_profiles.sh on the GPU box) into the small,
committed summaries under profiles/: the rocprofv3 --stats kernel table and the per-launch HBM
traffic of every kernel from the FETCH_SIZE / WRITE_SIZE PMC passes.

Units/corrections (MI355X_MICROARCH.md §HBM): counters are KiB; on gfx950 FETCH_SIZE reports 1/2 of
the bytes of wide coalesced reads. Calibrated in the same run on a kernel with a known byte count
(head_fwd_kernel of gpt_block.hip: reads x = N*16*L*4 B, writes qkv = N*48*L*4 B, nothing else of
size): read factor and write factor are stored next to the numbers."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r01")
dst = os.path.join(ROOT, "profiles")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "r01_kernel_stats.csv"), os.path.join(dst, "r01_kernel_stats.csv"))
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))


def per_launch(tag, counter):
    return {k: v[counter]["sum"] / v[counter]["dispatches"] for k, v in pmc[tag].items() if counter in v}


fetch, write = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
add = [k for k in fetch if "head_fwd_kernel" in k][0]
n_px = batch * 784  # pixels per launch
read_factor = (n_px * 16 * 4) / (fetch[add] * 1024)
write_factor = (n_px * 48 * 4) / (write[add] * 1024)
table = {}
for k in fetch:
    table[k] = {
        "fetch_kib_raw": fetch[k], "write_kib_raw": write.get(k),
        "hbm_read_bytes": fetch[k] * 1024 * read_factor,
        "hbm_write_bytes": (write.get(k) or 0.0) * 1024 * write_factor,
    }
dkv = [k for k in table if ("attn_dkv_m44_kernel" in k or "attn_bwd_dkv_kernel" in k)][0]
out = {
    "per_gpu_batch": batch,
    "calibration": {"kernel": add, "read_factor": read_factor, "write_factor": write_factor},
    "attn_bwd_dkv_bytes_per_launch": table[dkv]["hbm_read_bytes"] + table[dkv]["hbm_write_bytes"],
    "kernels": table,
}
json.dump(out, open(os.path.join(dst, "r01_traffic.json"), "w"), indent=1)
sq = pmc.get("sq", {})
json.dump({k: {c: v[c]["sum"] / v[c]["dispatches"] for c in v} for k, v in sq.items() if "attn" in k},
          open(os.path.join(dst, "r01_attention_sq_counters.json"), "w"), indent=1)
print("read_factor %.3f write_factor %.3f" % (read_factor, write_factor))
for k in table:
    if "attn" in k or "conv" in k or "ln_" in k:
        print("%-70s read %8.1f MB write %8.1f MB" % (k.split("(")[0][-70:], table[k]["hbm_read_bytes"] / 1e6,
                                                      table[k]["hbm_write_bytes"] / 1e6))
