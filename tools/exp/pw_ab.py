"""Forward time of 1x1 convolutions on the bf16x3 kernels (run once per setting of PG_CONV_B3_PW: the
switch is read once per process). usage: PG_CONV_B3_PW={0,1} python tools/exp/pw_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch
from pytorch_generative_amd import nn as pg_nn

dev = torch.device("cuda:0")
torch.manual_seed(0)
CASES = [  # cin, cout, (n, h, w), forward kwargs
    (64, 64, (512, 32, 32), dict(in_act="elu", out_act="elu")),
    (64, 64, (512, 32, 32), {}),
    (32, 64, (512, 32, 32), dict(in_act="elu", out_act="elu")),
    (64, 32, (1024, 28, 28), dict(in_act="relu")),
    (32, 64, (1024, 28, 28), dict(in_act="relu")),
    (128, 128, (512, 32, 32), {}),
    (128, 256, (256, 32, 32), {}),
]
for cin, cout, (n, h, w), kw in CASES:
    conv = pg_nn.Conv2d(cin, cout, 1).to(dev)
    x = torch.randn(n, cin, h, w, device=dev)
    with torch.no_grad():
        for _ in range(3):
            conv(x, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            conv(x, **kw)
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 * 1e6
    mb = n * h * w * (cin + cout) * 4 / 1e6
    gf = 2.0 * n * h * w * cin * cout / 1e9
    print(f"1x1 {cin:3d}->{cout:3d} N={n} {h}x{w} {kw}: {us:7.1f} us  {mb / us:6.2f} TB/s(alg)  {gf / us * 1e3:6.1f} TF/s", flush=True)
