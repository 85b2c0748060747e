"""Weight-gradient kernel timing and error against an fp64 reference, straight through the C-ABI.
Run twice to A/B the bf16x3 kernel:  PG_WGRAD_B3=0 python tools/exp/wgrad_ab.py ; python tools/exp/wgrad_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch
import torch.nn.functional as F
from pytorch_generative_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
ACT = {None: lambda t: t, "relu": F.relu, "elu": F.elu, "gelu": F.gelu}
CASES = [
    # name, N, Cin, H, W, Cout, kh, kw, ph, pw, act
    ("snail 2x2 64->64 N512", 512, 64, 32, 32, 64, 2, 2, 1, 1, "elu"),
    ("snail 2x2 64->128 N512", 512, 64, 32, 32, 128, 2, 2, 1, 1, "elu"),
    ("snail 2x2 64->64 N128", 128, 64, 32, 32, 64, 2, 2, 1, 1, "elu"),
    ("gated 1x3 128->256 N128", 128, 128, 32, 32, 256, 1, 3, 0, 1, None),
    ("gated 2x1 256->256 N128", 128, 256, 32, 32, 256, 2, 1, 2, 0, None),
    ("3x3 64->64 N64 relu", 64, 64, 32, 32, 64, 3, 3, 1, 1, "relu"),
    ("2x3 32->64 16x16", 32, 32, 16, 16, 64, 2, 3, 1, 1, "gelu"),
]
for name, n, cin, h, w, cout, kh, kw, ph, pw, act in CASES:
    torch.manual_seed(0)
    x = torch.randn(n, cin, h, w, device=dev)
    dy = torch.randn(n, cout, h, w, device=dev)
    spec = ops.ConvSpec(kh, kw, ph, pw)
    T = len(spec.wg_taps)
    dw = torch.zeros(cout, cin, kh, kw, device=dev)
    db = torch.zeros(cout, device=dev)
    ws_n = lib.pg_conv2d_wgrad_workspace_floats(cout, cin, T)
    ws = torch.empty(ws_n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        return lib.pg_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), n, cin, h, w,
                                   cout, h, w, kh, kw, T, spec.w_dr, spec.w_dc, spec.w_u, spec.w_v,
                                   ops._ACT_IDS[act], ws.data_ptr(), ws_n, st)
    _lib.check(call(), "wgrad")
    torch.cuda.synchronize()
    xa = ACT[act](x.double())
    xp = F.pad(xa, (pw, pw, ph, ph))
    ref = torch.zeros(cout, cin, kh, kw, device=dev, dtype=torch.float64)
    for u in range(kh):
        for v in range(kw):
            ref[:, :, u, v] = torch.einsum("nohw,nihw->oi", dy.double(), xp[:, :, u:u + h, v:v + w])
    ew = float((dw.double() - ref).abs().max() / ref.abs().max())
    eb = float((db.double() - dy.double().sum((0, 2, 3))).abs().max() / dy.double().sum((0, 2, 3)).abs().max())
    for _ in range(3): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): call()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
    gf = 2.0 * n * h * w * cin * cout * T / 1e9
    print(f"{name:28s} err dw {ew:.1e} db {eb:.1e} | {us:8.1f} us  {gf / us * 1e3:6.1f} TF/s", flush=True)
