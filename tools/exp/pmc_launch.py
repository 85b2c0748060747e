"""A few launches of each kernel the round's counter passes look at, through the C-ABI, for rocprofv3 (--kernel-trace /
--pmc passes are wrapped around THIS command by tools/collect_profiles_r05.sh):

    python tools/exp/pmc_launch.py [launches]

  * calibration: ops.add on (1024, 64, 32, 32) fp32 — reads 2 x 268.4 MB, writes 1 x 268.4 MB (the units of FETCH_SIZE /
    WRITE_SIZE are derived from it in the same process, MI355X_MICROARCH.md HBM section);
  * the headline's attention kernels at the bench's batch 1024 (ImageGPT: 4 heads, d_k = d_v = 4, L = 784): attn_fwd_m44,
    the fused backward attn_bwd_m44, and the two-kernel backward;
  * PixelSNAIL's dominant convolution (2x2 64 -> 64, ELU prologue) at batch 1024;
  * weight gradients: 2x2 64 -> 64 at batch 1024 (PixelSNAIL), 1x1 128 -> 256 and 2x3 128 -> 256 at batch 512 (GatedPixelCNN).
Prints the HIP-event time of every launch family as one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from pytorch_generative_amd import ops  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
dev = torch.device("cuda:0")
orig = bench._event_time
bench._event_time = lambda fn, stream, iters=10: orig(fn, stream, iters=launches)
a = torch.randn(1024, 64, 32, 32, device=dev)
b = torch.randn(1024, 64, 32, 32, device=dev)
for _ in range(launches):
    ops.add(a, b)
torch.cuda.synchronize()
del a, b
out = {"launches": launches}
r = bench.attention_kernel_roofline(1024, dev, 4, 4, 4, 28, False)
out["attn_m44_b1024"] = {k: round(v["launch_ms"], 4) for k, v in r.items()}
out["conv_2x2_64_64_b1024"] = round(bench.conv_kernel_roofline(1024, dev)["launch_ms"], 4)
out["wgrad_2x2_64_64_b1024"] = round(bench.wgrad_kernel_roofline(1024, dev)["launch_ms"], 4)
out["wgrad_1x1_128_256_b512"] = round(bench.wgrad_kernel_roofline(512, dev, 128, 256, 32, (1, 1, 0, 0))["launch_ms"], 4)
out["wgrad_2x3_128_256_b512"] = round(bench.wgrad_kernel_roofline(512, dev, 128, 256, 32, (2, 3, 1, 1))["launch_ms"], 4)
# GatedPixelCNN's widest 1x1 (256 -> 256, batch 512) on the wide kernel: a forward launch through the public op
if "--wide" in sys.argv:
    import torch.nn.functional as F  # noqa: F401
    xw = torch.randn(512, 256, 32, 32, device=dev)
    ww = (torch.randn(256, 256, 1, 1, device=dev) * 0.05)
    bw = torch.zeros(256, device=dev)
    spec = ops.ConvSpec(1, 1, 0, 0)
    with torch.no_grad():
        for _ in range(launches):
            ops.conv2d_taps(xw, ww, bw, spec, out_hw=(32, 32))
    torch.cuda.synchronize()
# round 6: the overlapped 16-wave kernel (conv_b3q_kernel, two tiles per workgroup) on the shape it is routed for — PixelCNN++'s 2x3
# 160 -> 320 at batch 64 — and, for comparison, the same launch count of GatedPixelCNN's 2x1 256 -> 256 on the wide kernel
if "--q" in sys.argv:
    for cin, cout, k, pad, batch in ((160, 320, (2, 3), (1, 1), 64), (256, 256, (2, 1), (2, 0), 512)):
        xq = torch.randn(batch, cin, 32, 32, device=dev)
        wq = torch.randn(cout, cin, *k, device=dev) * 0.05
        bq = torch.zeros(cout, device=dev)
        spec = ops.ConvSpec(k[0], k[1], pad[0], pad[1])
        with torch.no_grad():
            for _ in range(launches):
                ops.conv2d_taps(xq, wq, bq, spec, out_hw=(32, 32))
        torch.cuda.synchronize()
print(json.dumps(out))
