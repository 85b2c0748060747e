"""Weight-gradient launches (pg_conv2d_wgrad through the C-ABI, HIP events) on the shapes of PixelSNAIL / GatedPixelCNN / PixelCNN++:
one JSON line per shape. A/B: once with the production library, once with PG_HIP_LIB=<lib/libpg_hip_ab.so> and a switch
(PG_WGRAD_B3_RING=0: the row-ring kernel off)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
CASES = [
    ("snail 2x2 64->64 b1024", 1024, 64, 64, 32, (2, 2, 1, 1)),
    ("snail 2x2 64->128 b1024", 1024, 64, 128, 32, (2, 2, 1, 1)),
    ("snail 2x2 64->64 b128", 128, 64, 64, 32, (2, 2, 1, 1)),
    ("gated 2x1 256->256 b512", 512, 256, 256, 32, (2, 1, 2, 0)),
    ("gated 1x3 128->256 b512", 512, 128, 256, 32, (1, 3, 0, 1)),
    ("gated 1x2 128->256 b512", 512, 128, 256, 32, (1, 2, 0, 1)),
    ("gated 2x3 128->256 b512", 512, 128, 256, 32, (2, 3, 1, 1)),
    ("gated 1x1 128->256 b512", 512, 128, 256, 32, (1, 1, 0, 0)),
    ("gated 1x1 256->256 b512", 512, 256, 256, 32, (1, 1, 0, 0)),
    ("snail 1x1 64->64 b1024", 1024, 64, 64, 32, (1, 1, 0, 0)),
    ("gated 2x1 128->256 b512", 512, 128, 256, 32, (2, 1, 2, 0)),
    ("pcnnpp 2x2 320->320 16x16 b64", 64, 320, 320, 16, (2, 2, 1, 1)),
]
SEL = sys.argv[1:]
for name, batch, cin, cout, hw, k in CASES:
    if SEL and not any(s in name for s in SEL):
        continue
    ts = sorted(bench.wgrad_kernel_roofline(batch, dev, cin, cout, hw, k)["launch_ms"] for _ in range(3))
    r = bench.wgrad_kernel_roofline(batch, dev, cin, cout, hw, k)
    print(json.dumps({"case": name, "launch_ms": round(ts[1], 4), "tflops": round(r["flop_per_launch"] / ts[1] / 1e9, 1)}), flush=True)
