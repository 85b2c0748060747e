"""The reference suite's PixelSNAIL reproduce() case (batch 1, 28x28, x ~ N(0, 1)) under the strict canary allocator, with the
step graphed (as the test runs it) and eager: which parameters end up non-finite, and does the eager step show it too?"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd"), os.path.join(ROOT, "tests")]
import guard  # noqa: E402

if guard.enabled():
    guard.install()
import torch  # noqa: E402

import pytorch_generative_amd as pg  # noqa: E402
from pytorch_generative_amd import ops, optim, recipes, trainer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn((1, 1, 28, 28))


def build():
    torch.manual_seed(1)
    return pg.models.PixelSNAIL(in_channels=1, out_channels=1, n_channels=64, n_pixel_snail_blocks=8, n_residual_blocks=2,
                                attention_value_channels=32, attention_key_channels=4).to(dev)


for graph in (False, True, True):
    model = build()
    opt = optim.FlatAdam(model.parameters(), lr=1e-3, lr_decay=0.999977)

    class L:
        def __iter__(self):
            return iter([(x, torch.tensor([0]))])

    with tempfile.TemporaryDirectory() as d:
        t = trainer.Trainer(model=model, loss_fn=recipes.bce_loss, optimizer=opt, train_loader=L(), eval_loader=L(),
                            log_dir=d, n_gpus=0, graph=graph)
        t.interleaved_train_and_eval(1)
    torch.cuda.synchronize()
    bad = [(n, int((~torch.isfinite(p)).sum()), p.numel()) for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
    print(f"graph={graph}: {len(bad)} parameters with non-finite values", bad[:8])
    g = opt.flat_grad
    print("   flat grad non-finite:", int((~torch.isfinite(g)).sum()), "of", g.numel(), " state block", opt.state_block.tolist()[:6])
    big = sorted(((float(p._pg_grad.abs().max()), n) for n, p in model.named_parameters()), reverse=True)[:6]
    print("   largest |grad| per parameter:", [(f"{v:.3g}", n) for v, n in big])
