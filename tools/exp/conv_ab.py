"""A/B of the convolution kernels on the GPU box: MFMA implicit GEMM vs VALU tap kernel — values
and time per shape (forward and data gradient), through the autograd ops.
usage: python tools/exp/conv_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch
from pytorch_generative_amd import ops
from pytorch_generative_amd import nn as pg_nn

dev = torch.device("cuda:0")
torch.manual_seed(0)

def run(conv, x, kw, mfma):
    ops.conv.CONV_MFMA = mfma  # (the flag lives in the ops.conv module since the round-6 split)
    x = x.clone().requires_grad_(True)
    y = conv(x, **kw)
    g = torch.ones_like(y) * 0.5 + torch.arange(y.numel(), device=dev).reshape(y.shape).remainder(7) * 0.1
    y.backward(g)
    return y.detach(), x.grad.detach(), conv.weight.grad.detach().clone()

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6

CASES = [
    ("big snail 2x2 64->64 N512", lambda: pg_nn.Conv2d(64, 64, 2, padding=1), (512, 64, 32, 32), dict(crop=(32, 32), in_act="elu")),
    # name, ctor, input shape, forward kwargs
    ("snail 2x2 64->64", lambda: pg_nn.Conv2d(64, 64, 2, padding=1), (128, 64, 32, 32), dict(crop=(32, 32), in_act="elu")),
    ("snail 2x2 64->128", lambda: pg_nn.Conv2d(64, 128, 2, padding=1), (128, 64, 32, 32), dict(crop=(32, 32), in_act="elu")),
    ("1x1 64->64", lambda: pg_nn.Conv2d(64, 64, 1), (128, 64, 32, 32), {}),
    ("1x1 69->36", lambda: pg_nn.Conv2d(69, 36, 1), (128, 69, 32, 32), {}),
    ("1x1 32->64 28x28", lambda: pg_nn.Conv2d(32, 64, 1), (64, 32, 28, 28), dict(in_act="relu")),
    ("causal 3x3B 32->32 28x28", lambda: pg_nn.CausalConv2d(False, 32, 32, 3, padding=1), (64, 32, 28, 28), dict(in_act="relu")),
    ("gated 1x3 128->256", lambda: pg_nn.Conv2d(128, 256, (1, 3), padding=(0, 1)), (32, 128, 32, 32), {}),
    ("gated 2x1 256->256 crop", lambda: pg_nn.Conv2d(256, 256, (2, 1), padding=(2, 0)), (32, 256, 32, 32), dict(crop=(32, 32))),
    ("1x1 256->256", lambda: pg_nn.Conv2d(256, 256, 1), (32, 256, 32, 32), {}),
    ("3x3 64->32 64x64", lambda: pg_nn.Conv2d(64, 32, 3, padding=1), (16, 64, 64, 64), dict(in_act="relu")),
    ("3x3 32->32 8x8", lambda: pg_nn.Conv2d(32, 32, 3, padding=1), (64, 32, 8, 8), dict(in_act="gelu")),
    ("3x3 32->32 4x4", lambda: pg_nn.Conv2d(32, 32, 3, padding=1), (50, 32, 4, 4), dict(in_act="gelu")),
    ("1x1 32->64 2x2", lambda: pg_nn.Conv2d(32, 64, 1), (50, 32, 2, 2), dict(in_act="gelu")),
    ("1x1 64->32 1x1", lambda: pg_nn.Conv2d(64, 32, 1), (50, 64, 1, 1), dict(in_act="gelu")),
    ("3x3 16->24 30x30", lambda: pg_nn.Conv2d(16, 24, 3, padding=1), (5, 16, 30, 30), {}),
]
SEL = sys.argv[1:]
if os.environ.get("PG_CAL"):  # traffic calibration for PMC passes: add_kernel reads 2 x 134 MB, writes 134 MB
    ca, cb = torch.randn(512, 64, 32, 32, device=dev), torch.randn(512, 64, 32, 32, device=dev)
    for _ in range(5):
        ops.add(ca, cb)
    torch.cuda.synchronize()
for name, ctor, shape, kw in CASES:
    if SEL and not any(k in name for k in SEL):
        continue
    conv = ctor().to(dev)
    x = torch.randn(shape, device=dev)
    y0, dx0, dw0 = run(conv, x, kw, False); conv.weight.grad = None
    y1, dx1, dw1 = run(conv, x, kw, True); conv.weight.grad = None
    ey = float((y1 - y0).abs().max() / y0.abs().max())
    ex = float((dx1 - dx0).abs().max() / dx0.abs().max())
    ew = float((dw1 - dw0).abs().max() / dw0.abs().max())
    def fwd(m):
        ops.conv.CONV_MFMA = m
        with torch.no_grad():
            conv(x, **kw)
    xg = x.clone().requires_grad_(True)
    def fb(m):
        ops.conv.CONV_MFMA = m
        y = conv(xg, **kw)
        y.backward(y0)
    t0, t1 = timeit(lambda: fwd(False)), timeit(lambda: fwd(True))
    b0, b1 = timeit(lambda: fb(False)), timeit(lambda: fb(True))
    n, cin, h, w = shape
    cout = conv.weight.shape[0]
    taps = len(conv._conv_spec().fwd_taps)
    gf = 2.0 * n * h * w * cin * cout * taps / 1e9
    print(f"{name:28s} err y {ey:.1e} dx {ex:.1e} dw {ew:.1e} | fwd {t0:7.1f} -> {t1:7.1f} us ({gf / t1 * 1e3:6.1f} TF/s) | fwd+bwd {b0:7.1f} -> {b1:7.1f} us", flush=True)
