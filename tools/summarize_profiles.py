"""Turns gpurun_out/prof_r03/ (made by tools/collect_profiles.sh on the GPU box) into the small,
committed summaries under profiles/: the rocprofv3 --stats kernel tables of both bench workloads,
per-launch HBM traffic of every kernel from the FETCH_SIZE / WRITE_SIZE PMC passes, and the SQ
counters of the dominant kernels.

Units/corrections (MI355X_MICROARCH.md §HBM): counters are KiB; on gfx950 FETCH_SIZE under-reports
wide coalesced reads. Calibrated in the same run on kernels with known byte counts: ImageGPT run —
head_fwd_kernel of gpt_block.hip (reads x = N*16*L*4 B, writes qkv = N*48*L*4 B); PixelSNAIL run —
gated_fwd4_kernel (reads 3*N*64*L*4 B, writes N*64*L*4 B). Factors are stored next to the numbers."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r03")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
RUNS = {
    "igpt": dict(batch=1024, cal="head_fwd_kernel", read_b=lambda n: n * 784 * 16 * 4, write_b=lambda n: n * 784 * 48 * 4,
                 dominant="attn_bwd_m44_kernel"),
    "snail": dict(batch=512, cal="gated_fwd4_kernel", read_b=lambda n: 3 * n * 64 * 1024 * 4,
                  write_b=lambda n: n * 64 * 1024 * 4, dominant="conv_b3_kernel<false, 4, 4, 1, false, false>"),
}


def per_launch(tag, counter):
    return {k: v[counter]["sum"] / v[counter]["dispatches"] for k, v in pmc.get(tag, {}).items() if counter in v}


for run, cfg in RUNS.items():
    shutil.copy(os.path.join(src, f"{run}_kernel_stats.csv"), os.path.join(dst, f"r03_{run}_kernel_stats.csv"))
    fetch, write = per_launch(f"{run}_fetch", "FETCH_SIZE"), per_launch(f"{run}_write", "WRITE_SIZE")
    if not fetch:  # PMC passes of this workload were not collected (see profiles/README.md)
        print(f"[{run}] kernel stats only")
        continue
    cal = [k for k in fetch if cfg["cal"] in k][0]
    rf = cfg["read_b"](cfg["batch"]) / (fetch[cal] * 1024)
    wf = cfg["write_b"](cfg["batch"]) / (write[cal] * 1024)
    table = {k: {"fetch_kib_raw": fetch[k], "write_kib_raw": write.get(k),
                 "hbm_read_bytes": fetch[k] * 1024 * rf, "hbm_write_bytes": (write.get(k) or 0.0) * 1024 * wf}
             for k in fetch}
    dom = [k for k in table if cfg["dominant"] in k][0]
    out = {"per_gpu_batch": cfg["batch"], "calibration": {"kernel": cal, "read_factor": rf, "write_factor": wf},
           "dominant_kernel": dom,
           "dominant_bytes_per_launch": table[dom]["hbm_read_bytes"] + table[dom]["hbm_write_bytes"],
           "kernels": table}
    json.dump(out, open(os.path.join(dst, "r03_traffic.json" if run == "igpt" else f"r03_{run}_traffic.json"), "w"), indent=1)
    sq = pmc.get(f"{run}_sq", {})
    keep = {k: {c: v[c]["sum"] / v[c]["dispatches"] for c in v} for k, v in sq.items()
            if any(t in k for t in ("attn", "conv_b3", "conv_mfma", "conv_wgrad", "tail_bwd"))}
    for k, v in keep.items():
        # matrix-pipe busy fraction: busy SIMD-cycles over the kernel's SIMD-cycles; GRBM_GUI_ACTIVE is summed
        # over the 8 XCDs and the chip has 1024 SIMDs (the normalisation of r02_snail_conv_b3_pmc.json)
        if v.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            v["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128.0)
    json.dump(keep, open(os.path.join(dst, f"r03_{run}_sq_counters.json"), "w"), indent=1)
    print(f"[{run}] read_factor {rf:.3f} write_factor {wf:.3f}; dominant {dom[:60]}: "
          f"{out['dominant_bytes_per_launch'] / 1e6:.1f} MB per launch")
