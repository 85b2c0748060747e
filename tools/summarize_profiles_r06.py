"""gpurun_out/prof_r06/pmc.json (tools/collect_profiles_r06.sh pmc, on the GPU box) -> the tracked round-6 counter records:

  profiles/r06_traffic.json      HBM bytes per launch + SQ counters of the headline's attention kernels at batch 1024
                                 (the file bench.py's roofline.traffic is read from)
  profiles/r06_wgrad_pmc.json    the same for the weight-gradient kernels (2x2 64->64 at batch 1024; 1x1 and 2x3 128->256 at 512)
  profiles/r06_snail_conv_pmc.json  PixelSNAIL's dominant convolution at batch 1024

Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are calibrated IN THE SAME PROCESS on add_kernel over
(1024, 64, 32, 32) fp32 (reads 2 x 268.4 MB, writes 268.4 MB) — on gfx950 FETCH_SIZE counts ~2 KiB per unit for wide coalesced
reads, WRITE_SIZE 1 KiB. mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): GRBM_GUI_ACTIVE is summed over the
8 XCDs, the chip has 1024 SIMDs."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r06", "pmc.json")
dst = os.path.join(ROOT, "profiles")
pmc = json.load(open(src))
per = {k: {c: v["per_dispatch"] for c, v in d.items()} for k, d in pmc.items()}
cal = [v for k, v in per.items() if "add_kernel" in k][0]
ACT = 1024 * 64 * 32 * 32 * 4
rb, wb = 2 * ACT / cal["FETCH_SIZE"], ACT / cal["WRITE_SIZE"]
calib = {"kernel": "add_kernel on (1024, 64, 32, 32) fp32, same process", "bytes_per_FETCH_SIZE_unit": rb,
         "bytes_per_WRITE_SIZE_unit": wb, "add_kernel": {"FETCH_SIZE": cal["FETCH_SIZE"], "WRITE_SIZE": cal["WRITE_SIZE"]}}


def rec(name, alg_read, alg_write, flop=None):
    k = [kk for kk in per if name in kk][0]
    v = per[k]
    out = {"hbm_read_bytes": v["FETCH_SIZE"] * rb, "hbm_write_bytes": v["WRITE_SIZE"] * wb,
           "algorithmic_read_bytes": alg_read, "algorithmic_write_bytes": alg_write,
           "traffic_over_algorithmic": (v["FETCH_SIZE"] * rb + v["WRITE_SIZE"] * wb) / (alg_read + alg_write),
           "SQ_INSTS_MFMA": v["SQ_INSTS_MFMA"], "SQ_INSTS_VALU": v["SQ_INSTS_VALU"],
           "SQ_VALU_MFMA_BUSY_CYCLES": v["SQ_VALU_MFMA_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": v["GRBM_GUI_ACTIVE"],
           "SQ_WAVE_CYCLES": v["SQ_WAVE_CYCLES"], "SQ_BUSY_CYCLES": v["SQ_BUSY_CYCLES"],
           "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128.0),
           # resident waves per SIMD while the kernel ran: wave-cycles over the kernel's SIMD-cycles; SQ_WAVE_CYCLES ticks once
           # per 4 cycles (x4 reproduces the nominal occupancies: 3.3-3.7 for the four-waves-per-SIMD kernels, 1.8 for <6, 2, 4>)
           "waves_per_simd": 4.0 * v["SQ_WAVE_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 128.0)}
    if flop:
        out["flop_per_launch"] = flop
    return k, out


N, H, L, D = 1024, 4, 784, 4
plane = N * H * D * L * 4      # one of q / k / v / o / dO / dq / dk / dv: 51.4 MB
lse = N * H * L * 4
pairs = N * H * L * (L + 1) / 2
attn = dict([rec("attn_fwd_m44_kernel", 3 * plane, plane + lse, pairs * 16),
             rec("attn_bwd_m44_kernel", 5 * plane + lse, 3 * plane, pairs * 40),
             rec("attn_dq_m44_kernel", 5 * plane + lse, plane + lse, pairs * 24),
             rec("attn_dkv_m44_kernel", 4 * plane + 2 * lse, 2 * plane, pairs * 32)])
json.dump({"what": "ImageGPT's attention kernels (4 heads, d_k = d_v = 4, L = 784) at the bench's batch 1024, 3 launches each "
                   "through the C-ABI (tools/exp/pmc_launch.py), separate rocprofv3 --pmc passes (tools/collect_profiles_r06.sh pmc), "
                   "round-6 library",
           "per_gpu_batch": 1024, "calibration": calib, "kernels": attn},
          open(os.path.join(dst, "r06_traffic.json"), "w"), indent=1)

act = lambda n, c: n * c * 32 * 32 * 4  # noqa: E731
wg = dict([rec("conv_wgrad_b3r_kernel<4", act(1024, 64) * 2, 64 * 64 * 4 * 4, 2.0 * 1024 * 1024 * 64 * 64 * 4),
           rec("conv_wgrad_b3r_kernel<1, 2, 4", act(512, 128) + act(512, 256), 128 * 256 * 4, 2.0 * 512 * 1024 * 128 * 256),
           rec("conv_wgrad_b3r_kernel<6", act(512, 128) + act(512, 256), 128 * 256 * 6 * 4, 2.0 * 512 * 1024 * 128 * 256 * 6)])
json.dump({"what": "bf16x3 weight-gradient kernels: 2x2 64->64 at batch 1024 (PixelSNAIL; the row-ring kernel conv_wgrad_b3r_kernel<4, 3> — the record of "
                   "conv_wgrad_b3_kernel<4, 2, 8, 2> / <1, 4, 8, 4> / <6, 2, 4, 2> they replaced is profiles/r06_wgrad_pmc_before_ring.json), 1x1 128->256 "
                   "(conv_wgrad_b3r_kernel<1, 2, 4, 2>: 128 x 128 tiles, two workgroups per CU) and 2x3 128->256 (conv_wgrad_b3r_kernel<6, 1, 2, 3>: 84 KB ring, one workgroup per CU) at batch 512, 32x32 images, 3 launches each (tools/exp/pmc_launch.py); "
                   "algorithmic bytes = x and dy read once; writes = the partial rows (reduced by wgrad_reduce_kernel)",
           "calibration": calib, "kernels": wg}, open(os.path.join(dst, "r06_wgrad_pmc.json"), "w"), indent=1)

cv = dict([rec("conv_b3p_kernel", act(1024, 64) + 1024 * 64 * 96, act(1024, 64), 2.0 * 1024 * 1024 * 64 * 64 * 4)])
json.dump({"what": "PixelSNAIL's dominant convolution (2x2 64->64, ELU prologue, forward) at batch 1024, round-6 library",
           "per_gpu_batch": 1024, "calibration": calib, "kernels": cv},
          open(os.path.join(dst, "r06_snail_conv_pmc.json"), "w"), indent=1)
# round 6: the overlapped 16-wave kernel on the shape it is routed for, and the wide kernel on GatedPixelCNN's 2x1 / 1x1 256 -> 256
qk = dict([rec("conv_b3q_kernel", act(64, 160), act(64, 320), 2.0 * 64 * 1024 * 160 * 320 * 6),  # five output chunks each stage x: what L2 does not catch shows as traffic
           rec("conv_b3_kernel<false, 4, 4, 2, false, false, false>", act(512, 256), act(512, 256), None)])
json.dump({"what": "conv_b3q_kernel (16 waves, two tiles per workgroup) on PixelCNN++'s 2x3 160->320 at batch 64, and the wide kernel's "
                   "launches of the same pass (GatedPixelCNN's 1x1 and 2x1 256->256 at batch 512 share the instantiation: per-dispatch "
                   "means over both), round-6 library; algorithmic bytes: x read once, out written once",
           "calibration": calib, "kernels": qk}, open(os.path.join(dst, "r06_conv_q_pmc.json"), "w"), indent=1)
for name, d in (("attention", attn), ("wgrad", wg), ("conv", cv), ("conv q", qk)):
    for k, v in d.items():
        print(f"{name:9s} {k[:60]:60s} read {v['hbm_read_bytes'] / 1e6:7.1f} MB write {v['hbm_write_bytes'] / 1e6:7.1f} MB "
              f"(x{v['traffic_over_algorithmic']:.2f} algorithmic)  mfma_busy {v['mfma_busy_frac']:.3f}  waves/SIMD {v['waves_per_simd']:.2f}")
