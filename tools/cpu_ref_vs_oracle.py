"""How the CPU baseline of bench.py (the oracle's restatement, `cpu_baseline.kind = "port"`) relates to the REAL
reference trainer step: both timed here, in the build container (the only place /root/reference exists), same model,
batch, thread count and data — ImageGPT (BASELINE configs[1]) and PixelSNAIL (configs[3]).

The reference side runs the reference's own modules through a restatement of Trainer._train_one_batch
(/root/reference/pytorch_generative/trainer.py:173-193: zero_grad, forward, loss, backward, clip_grad_norm_(1e50),
Adam.step, MultiplicativeLR.step) — the Trainer class itself needs TensorBoard and data loaders, the step does not.

    python tools/cpu_ref_vs_oracle.py [--steps 5] [--threads N]   ->  profiles/r04_cpu_reference_vs_oracle.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402
from torch import optim  # noqa: E402

REF = "/root/reference"


def ref_step_timer(model, x, lr, decay, loss_fn):
    opt = optim.Adam(model.parameters(), lr=lr)
    sched = optim.lr_scheduler.MultiplicativeLR(opt, lr_lambda=lambda _: decay)

    def step():
        model.train()
        opt.zero_grad()
        loss = loss_fn(x, model(x))
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e50)
        if True:
            opt.step()
            sched.step()
        return float(loss), float(norm)

    return step


def oracle_step_timer(forward, state0, x, lr, **kw):
    from oracle import train as otrain

    state = {k: v.clone() for k, v in state0.items()}
    opt_state = otrain.new_opt_state()

    def step():
        _, loss, grads = otrain.loss_and_grads(forward, state, x, **kw)
        otrain.adam_step_(state, grads, opt_state, lr=lr)
        return float(loss), 0.0

    return step


def timed(step, n):
    step()  # warm-up
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    return sum(ts) / n, min(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _ref  # loads /root/reference's package with its torchvision / tensorboard imports bypassed

    ref_models = _ref.load().models  # the reference itself
    import bench
    from oracle import models as omodels
    import torch.nn.functional as F

    def bce(x, preds):  # the reference recipes' loss (image_gpt.py:139-144): sum over pixels, mean over the batch
        b = x.shape[0]
        return F.binary_cross_entropy_with_logits(preds, x, reduction="none").reshape(b, -1).sum(1).mean()

    out = {"where": "build container (no GPU), torch " + torch.__version__, "threads": args.threads,
           "logical_cores": os.cpu_count(), "timed_steps": args.steps, "models": {}}
    for name, ctor, batch, fwd, kw in (
        ("image_gpt", lambda: ref_models.ImageGPT(**bench.WORKLOADS["image_gpt"]["kw"]), 32, omodels.image_gpt, dict(n_heads=4)),
        ("pixel_snail", lambda: ref_models.PixelSNAIL(**bench.WORKLOADS["pixel_snail"]["kw"]), 8, omodels.pixel_snail, {}),
    ):
        w = bench.WORKLOADS[name]
        torch.manual_seed(0)
        model = ctor()
        state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        x = bench.synthetic_batch(batch, 0, w["chw"])
        r_mean, r_min = timed(ref_step_timer(model, x, w["lr"], w["decay"], bce), args.steps)
        o_mean, o_min = timed(oracle_step_timer(fwd, state0, x, w["lr"], **kw), args.steps)
        out["models"][name] = {
            "batch": batch, "reference_ms_per_step": r_mean * 1e3, "oracle_ms_per_step": o_mean * 1e3,
            "reference_images_per_s": batch / r_mean, "oracle_images_per_s": batch / o_mean,
            "oracle_over_reference_step_time": o_mean / r_mean,
            "best_step_ms": {"reference": r_min * 1e3, "oracle": o_min * 1e3},
        }
        print(name, json.dumps(out["models"][name]), flush=True)
    path = os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_oracle.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
