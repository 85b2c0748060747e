"""The overlapped 16-wave convolution kernel (csrc/conv_b3q_kernel.h) on the shapes it was built for: values against torch-CPU
at a small batch, time per launch (forward under no_grad; forward + backward) at the bench batch. One JSON line per shape.
    python tools/exp/q_ab.py [substring filters...]            # the library _lib picks (PG_HIP_LIB=... for a variant)
A/B: run once with the production library and once with PG_HIP_LIB=<lib/libpg_hip_ab.so> PG_CONV_B3Q=0."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pytorch_generative_amd import nn as pg_nn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, iters=20, reps=5):
    """median over `reps` event-timed windows of `iters` calls (single windows show 10x outliers on a fresh box)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(ts)[len(ts) // 2]


# name, (cin, cout, kernel, padding), crop to the input size?, forward kwargs, image size, bench batch
CASES = [
    ("gated 1x1 256->256", (256, 256, 1, 0), False, {}, 32, 512),
    ("gated 1x1 128->256", (128, 256, 1, 0), False, {}, 32, 512),
    ("gated 1x1 128->128", (128, 128, 1, 0), False, {}, 32, 512),
    ("gated 2x1 256->256", (256, 256, (2, 1), (2, 0)), True, {}, 32, 512),
    ("gated 1x2 128->256", (128, 256, (1, 2), (0, 1)), True, {}, 32, 512),
    ("gated 1x3 128->256", (128, 256, (1, 3), (0, 1)), False, {}, 32, 512),
    ("snail 2x2 64->128 elu", (64, 128, 2, 1), True, dict(in_act="elu"), 32, 1024),
    ("snail 2x2 64->64 elu", (64, 64, 2, 1), True, dict(in_act="elu"), 32, 1024),
    ("pcnnpp 2x3 160->160", (160, 160, (2, 3), (1, 1)), True, {}, 32, 64),
    ("pcnnpp 2x3 320->160", (320, 160, (2, 3), (1, 1)), True, {}, 32, 64),
    ("pcnnpp 2x3 160->320", (160, 320, (2, 3), (1, 1)), True, {}, 32, 64),
    ("pcnnpp 2x2 320->320 16x16", (320, 320, 2, 1), True, {}, 16, 64),
    ("pcnnpp 2x3 320->160 16x16", (320, 160, (2, 3), (1, 1)), True, {}, 16, 64),
    ("vae 3x3 64->64 relu", (64, 64, 3, 1), False, dict(in_act="relu"), 32, 512),
]
SEL = sys.argv[1:]
for name, (cin, cout, k, pad), crop, kw, hw, batch in CASES:
    if SEL and not any(s in name for s in SEL):
        continue
    conv = pg_nn.Conv2d(cin, cout, k, padding=pad).to(dev)
    kwf = dict(kw, crop=(hw, hw)) if crop else dict(kw)
    # ---- values at batch 3 (odd: the two-tile mode's idle half) against torch-CPU
    xs = torch.randn(3, cin, hw, hw)
    xg = xs.to(dev).requires_grad_(True)
    y = conv(xg, **kwf)
    g = torch.randn(y.shape)
    y.backward(g.to(dev))
    xc = xs.clone().requires_grad_(True)
    act = {"elu": F.elu, "relu": F.relu, None: (lambda t: t)}[kw.get("in_act")]
    yc = F.conv2d(act(xc), conv.weight.detach().cpu(), conv.bias.detach().cpu(), padding=pad)
    if crop:
        yc = yc[:, :, :hw, :hw]
    yc.backward(g)
    ey = float((y.detach().cpu() - yc.detach()).abs().max() / yc.detach().abs().max())
    ex = float((xg.grad.cpu() - xc.grad).abs().max() / xc.grad.abs().max())
    # ---- time at the bench batch
    x = torch.randn(batch, cin, hw, hw, device=dev)
    xr = x.clone().requires_grad_(True)
    gy = torch.randn(batch, cout, hw, hw, device=dev)

    def fwd():
        with torch.no_grad():
            conv(x, **kwf)

    def fb():
        conv(xr, **kwf).backward(gy)

    t_f, t_fb = timeit(fwd), timeit(fb)
    taps = len(conv._conv_spec().fwd_taps)
    gf = 2.0 * batch * hw * hw * cin * cout * taps / 1e9
    print(json.dumps({"case": name, "batch": batch, "err_y": ey, "err_dx": ex, "fwd_us": t_f, "fwd_tflops": gf / t_f * 1e3,
                      "fwd_bwd_us": t_fb}), flush=True)
