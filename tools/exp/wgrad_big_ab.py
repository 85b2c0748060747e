"""Launch times of the weight-gradient shapes the big tiles of conv_wgrad_b3 take (run once per PG_WGRAD_B3_BIG setting):
    PG_WGRAD_B3_BIG=0 python tools/exp/wgrad_big_ab.py ; python tools/exp/wgrad_big_ab.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [("1x1 128->256 b512", 512, 128, 256, 32, (1, 1, 0, 0)), ("1x1 256->256 b512", 512, 256, 256, 32, (1, 1, 0, 0)),
          ("1x1 128->128 b512", 512, 128, 128, 32, (1, 1, 0, 0)), ("2x1 128->256 b512", 512, 128, 256, 32, (2, 1, 1, 0)),
          ("1x2 128->256 b512", 512, 128, 256, 32, (1, 2, 0, 1)), ("1x1 64->64 b1024", 1024, 64, 64, 32, (1, 1, 0, 0)),
          ("1x1 64->128 b1024", 1024, 64, 128, 32, (1, 1, 0, 0)), ("1x3 128->128 b512", 512, 128, 128, 32, (1, 3, 0, 1))]
out = {"PG_WGRAD_B3_BIG": os.environ.get("PG_WGRAD_B3_BIG", "1")}
for name, n, cin, cout, hw, k in SHAPES:
    r = bench.wgrad_kernel_roofline(n, dev, cin, cout, hw, k)
    out[name] = {"ms": round(r["launch_ms"], 4), "tflops": round(r["tflops"], 1)}
print(json.dumps(out))
