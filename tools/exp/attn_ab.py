"""Causal-attention kernels vs a materialised torch reference on the GPU box (values + time).
usage: python tools/exp/attn_ab.py [N]   (PG_ATTN_MFMA=0 selects the VALU kernels)"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch
from pytorch_generative_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)


def ref(q, kv, heads, e, vd, strict):
    n, _, h, w = q.shape
    L = h * w
    k, v = kv[:, :e], kv[:, e:]
    qh = q.view(n, heads, e // heads, L).transpose(2, 3)
    kh = k.reshape(n, heads, e // heads, L).transpose(2, 3)
    vh = v.reshape(n, heads, vd // heads, L).transpose(2, 3)
    s = qh @ kh.transpose(2, 3) / math.sqrt(e // heads)
    mask = torch.tril(torch.ones(L, L, device=q.device), diagonal=-int(strict)).bool()
    s = s.masked_fill(~mask, -float("inf"))
    p = torch.softmax(s, -1).masked_fill(~mask, 0.0)
    p = torch.nan_to_num(p, nan=0.0)
    return (p @ vh).transpose(2, 3).reshape(n, vd, h, w)


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for (n, heads, e, vd, hw, strict) in [(3, 1, 4, 32, 32, True), (2, 1, 4, 32, 8, True), (2, 2, 8, 32, 16, False),
                                       (2, 1, 4, 16, 16, True), (N, 1, 4, 32, 32, True)]:
    q = torch.randn(n, e, hw, hw, device=dev, requires_grad=True)
    kv = torch.randn(n, e + vd, hw, hw, device=dev, requires_grad=True)
    o = ops.causal_attention(q, kv, heads, e, vd, strict)
    g = torch.randn_like(o)
    o.backward(g)
    dq, dkv = q.grad.clone(), kv.grad.clone()
    msg = ""
    if n <= 4:
        q.grad = kv.grad = None
        o_r = ref(q, kv, heads, e, vd, strict)
        o_r.backward(g)
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        msg = f"err o {rel(o, o_r):.1e} dq {rel(dq, q.grad):.1e} dk {rel(dkv[:, :e], kv.grad[:, :e]):.1e} dv {rel(dkv[:, e:], kv.grad[:, e:]):.1e}"
    with torch.no_grad():
        tf = timeit(lambda: ops.causal_attention(q, kv, heads, e, vd, strict))
    def fb():
        q.grad = kv.grad = None
        ops.causal_attention(q, kv, heads, e, vd, strict).backward(g)
    tb = timeit(fb)
    print(f"n={n} heads={heads} dk={e // heads} dv={vd // heads} L={hw * hw} strict={int(strict)} {msg} | fwd {tf:.1f} us, fwd+bwd {tb:.1f} us", flush=True)
