"""Micro-benchmark of the causal-attention kernels through the C-ABI (HIP events, random data).
usage: python tools/attn_bench.py [batch] [heads] [dk] [dv] [H] [W] [strict] [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-generative_amd"))
import torch  # noqa: E402

from pytorch_generative_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
batch, heads, dk, dv, H, W, strict, iters = (a + [512, 4, 4, 4, 28, 28, 0, 5][len(a):])
dev = torch.device("cuda:0")
e, v, L = heads * dk, heads * dv, H * W
g = torch.Generator().manual_seed(0)
q = torch.randn(batch, e, H, W, generator=g).to(dev).requires_grad_(True)
kv = torch.randn(batch, e + v, H, W, generator=g).to(dev).requires_grad_(True)
d_o = torch.randn(batch, v, H, W, generator=g).to(dev)
o = ops.causal_attention(q, kv, heads, e, v, bool(strict))
o.backward(d_o)
torch.cuda.synchronize()
tf = tb = 0.0
for _ in range(iters):
    s, m, t = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    s.record()
    o = ops.causal_attention(q, kv, heads, e, v, bool(strict))
    m.record()
    o.backward(d_o)
    t.record()
    torch.cuda.synchronize()
    tf += s.elapsed_time(m)
    tb += m.elapsed_time(t)
pairs = batch * heads * L * (L + 1) / 2
print(f"fwd {tf/iters:.3f} ms  bwd {tb/iters:.3f} ms  pairs {pairs:.3e}  "
      f"fwd {pairs*(2*dk+2*dv)/(tf/iters)/1e9:.1f} TF  bwd {pairs*(8*dk+6*dv)/(tb/iters)/1e9:.1f} TF")
