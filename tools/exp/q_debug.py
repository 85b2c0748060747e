"""Diagnostics for conv_b3q_kernel: structured inputs whose outputs say WHICH index mapping is off."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from pytorch_generative_amd import nn as pg_nn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
cin, cout, hw, n = int(os.environ.get("CIN", 32)), int(os.environ.get("COUT", 128)), 16, int(os.environ.get("NB", 2))
k, pad = (1, 1), (0, 0)
if os.environ.get("K") == "1x2":
    k, pad = (1, 2), (0, 1)
conv = pg_nn.Conv2d(cin, cout, k, padding=pad).to(dev)


def run(x, w, b, label):
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
        y = conv(x.to(dev), crop=(hw, hw) if k != (1, 1) else None).cpu()
    yc = F.conv2d(x, w, b, padding=pad)[:, :, :hw, :hw]
    bad = (y - yc).abs() > 1e-3 * yc.abs().max().clamp_min(1e-6)
    print(f"--- {label}: {int(bad.sum())} of {bad.numel()} wrong; max err {float((y - yc).abs().max()):.3e}")
    if bad.any():
        per_img = bad.flatten(1).sum(1).tolist()
        per_chunk16 = bad.reshape(n, cout // 16, 16, -1).sum((0, 2, 3)).tolist()
        per_px32 = bad.reshape(n, cout, -1, 32).sum((0, 1, 3)).tolist()
        print("   per image", per_img)
        print("   per 16-channel tile", per_chunk16)
        print("   per 32-pixel slice", per_px32)
        idx = bad.nonzero()[:6].tolist()
        for i in idx:
            print("   e.g.", i, "got", float(y[tuple(i)]), "want", float(yc[tuple(i)]))
    return y, yc


w0 = torch.zeros_like(conv.weight).cpu()
b = torch.arange(cout).float() + 1
x1 = torch.ones(n, cin, hw, hw)
run(x1, w0, b, "w = 0: out = bias[co] (epilogue channel mapping)")
w1 = w0.clone()
w1[:, 0, 0, -1] = 1.0  # last tap = the pixel itself for the 1x2 (pad 1, cropped) window
xp = torch.zeros(n, cin, hw, hw)
xp[:, 0] = torch.arange(hw * hw).float().reshape(hw, hw) + 1000 * torch.arange(n).float().reshape(n, 1, 1)
run(xp, w1, torch.zeros(cout), "w = delta(ci 0): out = pixel index + 1000 n (pixel / image mapping)")
w2 = w0.clone()
for ci in range(cin):
    w2[:, ci, 0, -1] = 0.0
xc = torch.zeros(n, cin, hw, hw)
for ci in range(cin):
    xc[:, ci] = ci + 1
w2[:, :, 0, -1] = torch.eye(cout, cin) if cout <= cin else torch.cat([torch.eye(cin)] * (cout // cin))
run(xc, w2, torch.zeros(cout), "w = identity over channels: out[co] = (co % cin) + 1 (K mapping)")
run(torch.randn(n, cin, hw, hw), torch.randn_like(w0) * 0.1, torch.randn(cout), "random")
w3 = w0.clone()
w3[:, 0, 0, -1] = torch.arange(cout).float() + 1
y, yc = run(x1, w3, torch.zeros(cout), "x = 1, w[co, 0] = co + 1: out[co] = co + 1 (A rows / epilogue channel mapping, no dependence on x channels)")
print("   got ", y[0, :20, 5, 5].tolist())
y, yc = run(xc, w2, torch.zeros(cout), "identity again")
print("   got ", y[0, :20, 5, 5].tolist())
print("   want", yc[0, :20, 5, 5].tolist())
