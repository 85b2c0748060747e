"""Every bench workload: K training steps launched eagerly against the same K steps replayed from the captured hipGraph.

    python tools/exp/graph_vs_eager_all.py [--steps 6] [--batch 8] [models...]

Round 4 found a node of a captured step (a hipMemset2DAsync) that was right on the first replay and wrong on every later one; the
graph-equals-eager TEST of the tier covers ImageGPT only. This sweeps all seven workloads at a small batch with the bit-reproducible
kernels (ops.set_deterministic): parameters after K steps must agree to round-off (the VAE families draw their noise from torch's
generator, reseeded identically for both runs). Exit code 1 on a mismatch. Not part of the test tier until it has run on a GPU once
(tools/exp/next_round.sh runs it)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

import bench  # noqa: E402  (workload table, synthetic batches)
import pytorch_generative_amd as pg  # noqa: E402
from pytorch_generative_amd import graph, ops, optim  # noqa: E402


def loss_of(name, w):
    if name in ("beta_vae", "vd_vae"):
        def f(xx, preds):
            recon, klm = ops.elbo_terms(preds[0], xx, preds[1])
            return recon + klm
        return f
    if name == "pixel_cnn_pp":
        return lambda xx, preds: ops.dmol_loss_sum_mean(preds, xx, w["kw"]["n_mix"])
    return lambda xx, preds: ops.bce_with_logits_sum_mean(preds, xx)


def run(name, batch, steps, dev, graphed):
    w = bench.WORKLOADS[name]
    torch.manual_seed(0)
    model = getattr(pg.models, w["ctor"])(**w["kw"]).to(dev)
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=w["lr"], lr_decay=w["decay"])
    xs = [bench.synthetic_batch(batch, 100 + i, w["chw"]).to(dev) for i in range(steps)]
    if name == "pixel_cnn_pp":
        xs = [x * 2.0 - 1.0 for x in xs]
    loss_fn = loss_of(name, w)
    losses = []
    if graphed:
        step = graph.GraphedTrainStep(model, opt, loss_fn, xs[0], warmup_iters=2, preserve_state=True)
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        for x in xs:
            losses.append(float(step(x)))
    else:
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        for x in xs:
            opt.zero_grad()
            loss = loss_fn(x, model(x))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return losses, {k: p.detach().clone() for k, p in model.named_parameters()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("models", nargs="*", default=[m for m in bench.WORKLOADS])
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    was = ops.set_deterministic(True)
    bad = 0
    for name in a.models:
        le, pe = run(name, a.batch, a.steps, dev, False)
        lg, pg_ = run(name, a.batch, a.steps, dev, True)
        noise = name in ("beta_vae", "vd_vae")  # graph replay advances the captured generator differently: losses only roughly
        worst, wk = 0.0, ""
        for k in pe:
            d = float((pe[k] - pg_[k]).abs().max()) / (float(pe[k].abs().max()) + 1e-12)
            if d > worst:
                worst, wk = d, k
        lerr = max(abs(x - y) / (abs(x) + 1e-12) for x, y in zip(le, lg))
        tol = 0.2 if noise else 1e-4  # noise families: the warm-up steps of the capture consume generator state, so the two runs
        # see different noise — only a gross failure (garbage, non-finite values) shows there
        ok = worst <= tol and lerr <= tol and all(l == l for l in lg)
        bad += not ok
        print(f"{name:16s} {'ok ' if ok else 'MISMATCH'} worst parameter difference {worst:.2e} ({wk}); worst loss difference {lerr:.2e}; "
              f"losses eager {le[0]:.4f} .. {le[-1]:.4f}, graph {lg[0]:.4f} .. {lg[-1]:.4f}" + ("  [noise-driven family: loose bound]" if noise else ""))
    ops.set_deterministic(was)
    print("all workloads agree" if not bad else f"{bad} workload(s) differ")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
