"""Which module's output / gradient first turns non-finite under the strict canary allocator (PG_GUARD=1 PG_GUARD_ALIGN=16)?
One eager training step of a model with forward hooks on every leaf module and gradient hooks on their outputs.
usage: PG_GUARD=1 PG_GUARD_ALIGN=16 AMD_SERIALIZE_KERNEL=3 python tools/exp/nan_hunt.py pixel_snail 1 28 1"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd"), os.path.join(ROOT, "tests")]
import guard  # noqa: E402

if guard.enabled():
    guard.install()
import torch  # noqa: E402

import pytorch_generative_amd as pg  # noqa: E402
from pytorch_generative_amd import ops, optim  # noqa: E402

name, batch, size, ch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda:0")
torch.manual_seed(0)
ctor = {"pixel_snail": lambda: pg.models.PixelSNAIL(in_channels=ch, out_channels=ch, n_channels=64, n_pixel_snail_blocks=8,
                                                   n_residual_blocks=2, attention_value_channels=32, attention_key_channels=4),
        "gated_pixel_cnn": lambda: pg.models.GatedPixelCNN(in_channels=ch, out_channels=ch, n_gated=10, gated_channels=128,
                                                           head_channels=32)}[name]
model = ctor().to(dev)
opt = optim.FlatAdam(model.parameters(), lr=1e-3)
bad = []
LIM = float(os.environ.get('PG_HUNT_LIMIT', '1e12'))


def _ok(t):
    return bool((torch.isfinite(t) & (t.abs() < LIM)).all())


def _nbad(t):
    return (~(torch.isfinite(t) & (t.abs() < LIM))).sum()


def fwd_hook(mod_name):
    def hook(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            if torch.is_tensor(o) and o.is_floating_point():
                if not bool(_ok(o)):
                    bad.append(f"forward  {mod_name} ({type(mod).__name__}) output {i} {tuple(o.shape)}: "
                               f"{int(_nbad(o))} non-finite values")
                if o.requires_grad:
                    o.register_hook(lambda g, n=mod_name, m=mod, i=i: bad.append(
                        f"backward grad of {n} ({type(m).__name__}) output {i} {tuple(g.shape)}: "
                        f"{int(_nbad(g))} non-finite values") if not bool(_ok(g)) else None)
    return hook


for n, m in model.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_hook(fwd_hook(n))
x = torch.randn((batch, ch, size, size)).to(dev)
opt.zero_grad()
loss = ops.bce_with_logits_sum_mean(model(x), x)
print("loss", float(loss))
loss.backward()
for n, p in model.named_parameters():
    g = p.grad if p.grad is not None else getattr(p, "_pg_grad", None)
    if g is not None and not bool(_ok(g)):
        bad.append(f"parameter gradient {n} {tuple(g.shape)}: {int(_nbad(g))} non-finite values")
opt.step()
torch.cuda.synchronize()
nonfinite = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
print("non-finite parameters after the step:", nonfinite[:6], len(nonfinite))
print("\n".join(bad[:25]) if bad else "no non-finite tensor seen by the hooks")
